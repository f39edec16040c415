from .pipeline_wan_fun_control import WanFunControlPipeline, WanPipelineOutput, denoise_latents  # noqa: F401
