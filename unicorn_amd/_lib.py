"""ctypes binding of libunicorn_hip.so (the reference-side stub a maintainer would add)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libunicorn_hip.so")

c_f = C.c_void_p   # device pointers travel as void*
c_i = C.c_int


class ModelCfg(C.Structure):
    _fields_ = [("dims", C.c_int32 * 4), ("depths", C.c_int32 * 4), ("num_classes", C.c_int32), ("mask", C.c_int32),
                ("n_layer_att", C.c_int32), ("embed_dim", C.c_int32), ("up_rate", C.c_int32), ("d_rate", C.c_int32),
                ("precision", C.c_int32)]


# name -> (restype, argtypes); every symbol declared in include/unicorn_hip.h
PROTOS = {
    "uni_last_error": (C.c_char_p, []),
    "uni_version": (c_i, []),
    "uni_ctx_create": (C.c_void_p, [c_i, C.POINTER(ModelCfg)]),
    "uni_ctx_destroy": (None, [C.c_void_p]),
    "uni_ctx_load_param": (c_i, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), c_i]),
    "uni_weights_file_cfg": (c_i, [C.c_char_p, C.POINTER(ModelCfg)]),
    "uni_ctx_load_file": (c_i, [C.c_void_p, C.c_char_p, C.POINTER(c_i)]),
    "uni_ctx_finalize": (c_i, [C.c_void_p, C.POINTER(c_i)]),
    "uni_ctx_missing_name": (C.c_char_p, [C.c_void_p, c_i]),
    "uni_ctx_reserve": (c_i, [C.c_void_p, c_i, c_i, c_i]),
    "uni_ctx_set_check": (c_i, [C.c_void_p, c_i]),
    "uni_ctx_stats": (c_i, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "uni_prof_begin": (c_i, [C.c_void_p]),
    "uni_prof_end": (c_i, [C.c_void_p, C.POINTER(C.c_double)]),
    "uni_backbone_fpn": (c_i, [C.c_void_p, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f, C.c_void_p]),
    "uni_interaction": (c_i, [C.c_void_p, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, C.c_void_p]),
    "uni_upsample": (c_i, [C.c_void_p, c_f, c_i, c_i, c_i, c_f, C.c_void_p]),
    "uni_head": (c_i, [C.c_void_p, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, C.c_void_p]),
    "uni_head_objects": (c_i, [C.c_void_p, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, C.c_void_p]),
    "uni_pos_embed": (c_i, [C.c_void_p, c_i, c_i, c_f, C.c_void_p]),
    "uni_msda_fwd": (c_i, [c_f, C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                           C.c_void_p]),
    "uni_corr_workspace_bytes": (C.c_size_t, [c_i, c_i, c_i]),
    "uni_corr_softmax_pv": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uni_corr_workspace_bytes_batched": (C.c_size_t, [c_i, c_i, c_i, c_i]),
    "uni_corr_softmax_pv_batched": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uni_prior_pyramid": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, C.c_void_p]),
    "uni_label_map_s8": (c_i, [c_f, c_f, c_i, c_i, C.c_void_p]),
    "uni_letterbox": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_f, C.POINTER(C.c_double), C.c_void_p]),
    "uni_decode_outputs": (c_i, [c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    "uni_nms_workspace_bytes": (C.c_size_t, [c_i]),
    "uni_nms": (c_i, [c_f, c_f, c_i, C.c_float, c_f, c_f, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uni_postprocess_workspace_bytes": (C.c_size_t, [c_i]),
    "uni_postprocess": (c_i, [c_f, c_i, c_i, c_i, C.c_float, C.c_float, c_i, c_i, c_f, c_f, c_f, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uni_sample_embeddings": (c_i, [c_f, c_i, c_i, c_i, c_f, c_i, c_i, C.c_float, c_f, C.c_void_p]),
    "uni_condinst_masks": (c_i, [c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uni_condinst_masks_u8": (c_i, [c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_double, c_i, c_i, C.c_float, c_f, c_f, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "uni_mask_resize": (c_i, [c_f, c_i, c_i, c_i, C.c_double, c_i, c_i, C.c_float, c_f, c_f, C.c_void_p]),
    "uni_vos_merge": (c_i, [c_f, c_f, c_i, c_i, c_i, C.c_double, c_f, c_f, c_i, c_i, c_i, c_f, C.c_void_p]),
    "uni_mots_overlap_free": (c_i, [c_f, c_i, c_i, c_i, c_f, C.c_void_p]),
    "uni_rle_workspace_bytes": (C.c_size_t, [c_i, c_i, c_i, c_i]),
    "uni_rle_encode": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uni_pack_weight": (c_i, [C.c_void_p, c_i, c_i, c_i, c_i, C.c_void_p]),
    "uni_gemm_bf16": (c_i, [c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_i,
                            c_f, c_i, c_i, C.c_void_p]),
    "uni_cast_bf16": (c_i, [c_f, c_i, c_f, c_i, c_i, c_i, C.c_void_p]),
    "uni_pack_weight_h2": (c_i, [C.c_void_p, c_i, c_i, c_i, c_i, C.c_void_p, C.POINTER(C.c_float)]),
    "uni_gemm_h2": (c_i, [c_f, c_i, c_f, C.c_float, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_i, c_f, c_i,
                          c_f, c_i, c_f, c_i, c_i, C.c_void_p]),
    "uni_cast_h2": (c_i, [c_f, c_i, c_f, c_i, c_i, c_i, C.c_void_p]),
    "uni_mlp_blob_bytes": (C.c_size_t, [c_i]),
    "uni_mlp_pack": (c_i, [C.c_void_p, C.c_void_p, C.c_void_p, c_i, c_i, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "uni_mlp_fused": (c_i, [c_f, c_i, c_f, c_f, c_f, C.c_float, C.c_float, c_f, c_i, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    "uni_layernorm": (c_i, [c_f, c_i, c_f, c_f, C.c_float, c_i, c_i, c_f, c_f, C.c_void_p]),
    "uni_dwconv7_ln": (c_i, [c_f, c_f, c_f, c_f, c_f, C.c_float, c_i, c_i, c_i, c_f, C.c_void_p]),
    "uni_dwconv7_ln_ex": (c_i, [c_f, c_f, c_f, c_f, c_f, C.c_float, c_i, c_i, c_i, c_i, c_f, c_i, C.c_void_p]),
    "uni_msda_tokens": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f, C.c_void_p]),
    "uni_groupnorm_act": (c_i, [c_f, c_f, c_f, c_f, C.c_float, c_i, c_i, c_i, c_i, c_f, c_f, C.c_void_p]),
    "uni_stem": (c_i, [c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_f, C.c_void_p]),
}

_lib = None
BUILD_CHECK = None      # outcome of verify_manifest for the loaded library ("verified" / "no-sources" / "skipped")


class UnicornHipError(RuntimeError):
    pass


def verify_manifest(lib_name):
    """The libraries are git-ignored build products that travel to the GPU box as files: unicorn_amd/csrc/build.sh writes
    lib/build_manifest.json (sha256 of every source / header and of both libraries) and this check holds the library that is about to be
    loaded, and the sources next to it (when the tree has them), to it -- a library that was not built from this tree's sources raises
    instead of running stale kernels.  Returns "verified" / "no-sources" (deployment without csrc/).  UNI_SKIP_MANIFEST=1 turns the check into
    a warning (local experiments with hand-built objects)."""
    import hashlib
    import json
    import warnings

    def sha(p):
        with open(p, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()

    def fail(msg):
        msg = "%s -- rebuild with `python -c 'import __graft_entry__ as g; g.build()'`" % msg
        if os.environ.get("UNI_SKIP_MANIFEST"):
            warnings.warn(msg)
            return "skipped"
        raise UnicornHipError(msg)

    mpath = os.path.join(_HERE, "lib", "build_manifest.json")
    if not os.path.exists(mpath):
        return fail("unicorn_amd/lib/build_manifest.json is missing (library built by hand?)")
    man = json.load(open(mpath))
    lpath = os.path.join(_HERE, "lib", lib_name)
    if man.get("libs", {}).get(lib_name) != sha(lpath):
        return fail("%s is not the library build_manifest.json describes" % lib_name)
    csrc, inc = os.path.join(_HERE, "csrc"), os.path.join(os.path.dirname(_HERE), "include")
    if not os.path.isdir(csrc):
        return "no-sources"
    assoc_srcs = ("assoc.cpp", "unicorn_assoc.h")                  # the host-side association library is built from these two only
    for name, digest in man.get("sources", {}).items():
        if (name in assoc_srcs) != (lib_name == "libunicorn_assoc.so"):
            continue
        src = os.path.join(inc if name.startswith("unicorn_") and name.endswith(".h") else csrc, name)
        if not os.path.exists(src) or sha(src) != digest:
            return fail("%s changed since %s was built (stale library)" % (name, lib_name))
    have = {f for f in os.listdir(csrc) if f.endswith(".h")} | ({f for f in os.listdir(inc) if f.endswith(".h")} if os.path.isdir(inc) else set())
    new = sorted(have - set(man.get("sources", {})))
    if new:
        return fail("header(s) %s are newer than the build manifest" % ", ".join(new))
    return "verified"


def lib():
    """Load libunicorn_hip.so (built by __graft_entry__.build() / unicorn_amd/csrc/build.sh).  No fallback."""
    global _lib
    if _lib is None:
        # torch first: PyTorch-ROCm brings its own libamdhip64; loading ours before it puts two HIP runtimes into the process and the
        # second one sees no device ("no HIP device 0 (count 0)")
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise UnicornHipError(
                "libunicorn_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU/PyTorch fallback for the Unicorn hot path)" % LIB_PATH)
        global BUILD_CHECK
        BUILD_CHECK = verify_manifest("libunicorn_hip.so")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOS.items():
            fn = getattr(L, name)      # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().uni_last_error()
        raise UnicornHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())
