// Go 19x19 slab - placeholder until the kernel lands (returns "unbuilt game").
#include "common.h"
BoardSlab* elfhip_make_go19() { return nullptr; }
