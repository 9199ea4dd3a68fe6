from .unicorn_sot import UnicornSOTTrack  # noqa: F401
from .quasi_dense_embed_tracker import QuasiDenseEmbedTracker  # noqa: F401
from .byte_tracker import BYTETracker  # noqa: F401
from .unicorn_vos import UnicornVOSTrack  # noqa: F401
from .omni import ByteMOTFrame, DemoPredictor, OmniMOTFrame, OmniMOTSFrame  # noqa: F401
