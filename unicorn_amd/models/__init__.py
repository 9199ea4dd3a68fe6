from .unicorn import Unicorn, UnicornHead, UnicornHeadMask, DynamicMaskHead, MODEL_CONFIGS  # noqa: F401
