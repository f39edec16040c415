from .wan_transformer4d import (ContextCache, WanAttentionBlock, WanSelfAttention,  # noqa: F401
                                WanTransformer4DModel)
