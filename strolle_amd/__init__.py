"""strolle_amd — MI355X-native hot path of Patryk27/strolle (BVH traversal, ReSTIR DI/GI, SVGF).

The product is the C-ABI shared library `strolle_amd/csrc/libstrolle_hip.so`
(include/strolle_hip.h); this package is its Python host-side mirror of
`strolle::Engine` plus scene helpers for the benchmark scenes.
"""
from .api import (Buffer, Camera, OutputFormat, CameraMode, Engine, Instance, Light, Material, Mesh, PassBit, StrolleError, Sun,  # noqa: F401
                  look_at_transform, perspective_infinite_reverse_rh)
