"""MI355X-native ray-trace core for raytracer-rust scenes: host-side package.

Layout:
  abi.py        ctypes mirror of include/rtk.h
  host.py       object model mirroring the reference (Scene, Camera, Material, ...)
  loader.py     the reference's YAML scene schema
  synthetic.py  deterministic synthetic scenes (throughput / parity configurations)
  canvas.py     Canvas: framebuffer container, 8-bit cast and PNG output
  device.py     binding of the HIP library (csrc/librtk.so); fails loudly without it
  dist.py       multi-GPU tile partition + gather (torch.distributed / RCCL)
"""
from . import abi, host, loader, synthetic  # noqa: F401
from .host import (  # noqa: F401
    PI, Camera, Material, Matrix4x4, Plane, PointLight, Scene, Sphere, Texture, point, rgb, vec3, vec4,
    BLACK, WHITE, RED,
)
from .loader import PackedScene  # noqa: F401
from .canvas import Canvas  # noqa: F401
