"""CPU oracle for the page-segmentation hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker.  The product (page-segmentation_amd/) never imports it.

PARITY UNPINNED for the layer arithmetic: the reference has no tests / golden vectors and its
arithmetic (TensorFlow 2.5, OpenCV 4.5.5, scikit-image 0.17.2, ocr4all-pylib) is not installed
and not installable offline.  Pinned pieces: see tests/golden/.
"""
from .core import (  # noqa: F401
    lib_path, build, conv2d, deconv2x2, maxpool2, round_bf16, preprocess, argmax,
    same_pad, set_num_threads, num_threads,
)
from .models import (  # noqa: F401
    ARCHS, weight_specs, init_weights, forward, predict_single_data, to_bf16_weights,
)
from .postprocess import (  # noqa: F401
    vote_connected_component_class, add_bounding_boxes, generate_output_masks,
    otsu_threshold, compute_char_height_from_gray, nearest_resize,
)
