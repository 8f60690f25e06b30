"""lmx — MI355X-native per-frame feature extraction behind the yolo / sam3 / dinov3 services of
UBC-AWP/vision-sam3-yolo-lameless.  The arithmetic lives in ``liblmx.so`` (hand-written HIP for
gfx950, C-ABI in ``include/lmx.h``); this package is the Python host side: ctypes bindings (``kernels``), the
launch sequences of the three networks (``dino``, ``yolo``, ``sam``) and the re-stated service contracts
(``services``).  There is NO CPU fallback: importing ``lmx.kernels`` without the built library raises."""

__version__ = "0.1.0"
