"""Drop-in mirror of the reference's import path `ocr4all_pixel_classifier.lib.*`, backed by the
MI355X engine in ../csrc (libpseg.so) instead of TensorFlow / OpenCV / scikit-image."""
__version__ = "0.6.5+mi355x.r1"
