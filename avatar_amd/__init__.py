"""avatar_amd — MI355X-native SMPL-to-depth fitting engine (hot path of sxyu/avatar's AvatarOptimizer).

Product path: avatar_amd/csrc (hand-written HIP for gfx950 behind the C ABI in include/avt.h), mirrored for
Python in avatar_amd.api.  avatar_amd.synth is the synthetic-workload harness.  The CPU oracle lives in
oracle/ and is never imported from here.
"""
__all__ = ["api", "capi", "synth"]
