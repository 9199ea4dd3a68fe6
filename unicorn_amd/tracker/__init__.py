from .unicorn_sot import UnicornSOTTrack  # noqa: F401
