"""MI355X-native multi-scale deformable attention (MSDeformAttn) for unipercept.

The compute path is hand-written HIP for gfx950, reached through the C ABI declared in
``include/msda.h`` (``libmsda.so``).  There is no CPU or pure-PyTorch fallback on the product
path: importing the kernels without the built library raises.
"""

__version__ = "0.1.0"
