from .fused_adamw import FusedAdamW  # noqa: F401
