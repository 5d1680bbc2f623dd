"""elf_amd: MI355X-native self-play hot path (boards -> masks/planes -> MCTS -> policy/value net)
behind the C-ABI of include/elfhip.h.  See DESIGN.md."""
from ._lib import AMERICAN, RUSSIAN, UGOLKI, GO19, F32, BF16, BITS, ElfHipError, game_info, lib  # noqa: F401
