"""unidefense_amd — MI355X-native (gfx950) implementation of the UniDefense dual-space reconstruction
training step behind the reference's ``model.load_model`` / ``engine.get_engine`` surface.

The compute path is the hand-written HIP library ``libunidefense_hip.so`` (see include/unidefense_hip.h);
importing the operators without it raises ``unidefense_amd.lib.UDLibraryError``.
"""
__all__ = ["lib", "kernels", "tape", "model", "engine", "loss"]
