"""vln-imagine_amd: MI355X-native HAMT/DUET cross-modal transformer hot path.

Sub-packages
  csrc/   hand-written HIP kernels for gfx950 + the C-ABI (`include/vlni.h`)
  hamt/   drop-in `models.*` for VLN-HAMT/finetune_src (put `hamt/` on sys.path)
  duet/   drop-in `models.*` for VLN-DUET/map_nav_src (put `duet/` on sys.path)
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
HAMT_PATH = os.path.join(PKG_DIR, "hamt")
DUET_PATH = os.path.join(PKG_DIR, "duet")
