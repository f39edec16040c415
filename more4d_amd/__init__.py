"""more4d_amd — MI355X-native (gfx950) implementation of MoRe4D's 4D-STraG denoising hot path.
Host side mirrors MoRe4D/{models,pipeline,utils,dist}; compute is libmore4d_hip.so (include/more4d_hip.h)."""
__version__ = "0.1.0"
