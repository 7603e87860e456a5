"""mdt_policy_amd -- MI355X-native (gfx950) implementation of MDT's diffusion-transformer action-denoising path.

Layout (only what the hot path needs):
  csrc/      hand-written HIP kernels + the C ABI (include/mdt_hip.h) -> csrc/libmdt_hip.so
  _lib.py    ctypes binding of that library (fails loudly if it is missing)
  models/    host-side mirror of the reference's ``mdt.models`` operator API:
             edm_diffusion.score_wrappers.GCDenoiser, edm_diffusion.gc_sampling.sample_*/get_sigmas_*,
             networks.mdtv_transformer.MDTVTransformer, networks.mdt_transformer.MDTTransformer
  sharding.py  batch-sharded sampling over the GPUs of one node (RCCL all-gather of the sampled actions)
  synthetic.py / configs.py  deterministic synthetic weights/inputs and canonical configurations
"""
__version__ = "0.1.0"
