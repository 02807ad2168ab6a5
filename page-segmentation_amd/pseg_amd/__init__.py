"""pseg_amd -- ctypes binding of libpseg.so (MI355X / gfx950 engine) plus build and
synthetic-data helpers.  The drop-in mirror of the reference API lives next to this package
in `ocr4all_pixel_classifier/`."""
from .engine import (  # noqa: F401
    Engine, PsegError, lib, lib_path, device_count, cc_vote, bbox_fill, masks, otsu_char_height,
    ARCH_IDS, MODE_F32_EXACT, MODE_BF16, EXPORTED_SYMBOLS, eval_confusion, cc_label, cc_tables, pinned_empty, pinned_empty_pooled, pinned_copy,
)
from .build import build_library  # noqa: F401
