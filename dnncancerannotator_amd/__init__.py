"""MI355X-native (gfx950 HIP) train/evaluate engine behind the DNNCancerAnnotator `annotator` surface.

Host code is plain Python over the C ABI of libdnnca.so (include/dnnca.h); no PyTorch, no TensorFlow."""

__version__ = '0.1.0'
