from .boxes import postprocess, postprocess_inst, nms, batched_nms  # noqa: F401
