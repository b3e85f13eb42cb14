"""MI355X-native Newton-step hot path for Waiwera-style geothermal flow simulation.

The compute path is the HIP library built from waiwera_amd/csrc (C-ABI in include/waiwera_hip.h).
There is no CPU fallback: importing the bindings without the built library raises.
"""
__version__ = "0.1.0"
