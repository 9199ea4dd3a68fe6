"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/unicorn_hip.h declares; host-side packing helpers agree with numpy; loud failure without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "unicorn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uni_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from unicorn_amd import _lib
    lib = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libunicorn_hip.so does not export %s" % s
    assert set(syms) == set(_lib.PROTOS), set(syms) ^ set(_lib.PROTOS)
    assert lib.uni_version() == 1


def test_pack_weight_matches_numpy():
    from unicorn_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    w = torch.randn(5, 16, 3, 3, generator=g)
    out = np.zeros((256, 192), dtype=np.uint16)
    wc = np.ascontiguousarray(w.numpy())
    assert lib.uni_pack_weight(wc.ctypes.data_as(C.c_void_p), 5, 16, 3, 3, out.ctypes.data_as(C.c_void_p)) == 0
    exp = w.permute(0, 2, 3, 1).reshape(5, 144).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(out[:5, :144], exp)
    assert not out[5:].any() and not out[:, 144:].any()


def test_pack_weight_h2_matches_numpy():
    """The host packer of the f16x2 operand format (the headline precision): per 8 consecutive k a 32-byte group
    [8 x hi][8 x lo] with hi = f16(x * scale), lo = f16(x * scale - hi), one power-of-two scale per tensor that puts max |w|
    into [2^14, 2^15); k = (ky * KW + kx) * Cin + c; rows / columns beyond N / K are zero."""
    from unicorn_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(1)
    N, Cin, k = 5, 16, 3
    w = torch.randn(N, Cin, k, k, generator=g) * 0.03
    w[0, 0, 0, 0] = 0.0
    w[1, 2, 1, 1] = 1e-7                                   # its lo half is an f16 subnormal / zero: must not blow up
    K = Cin * k * k
    Npad, Kpad = 256, (K + 63) // 64 * 64
    out = np.zeros((Npad, Kpad), dtype=np.uint32)
    wc = np.ascontiguousarray(w.numpy())
    sc = C.c_float(0)
    assert lib.uni_pack_weight_h2(wc.ctypes.data_as(C.c_void_p), N, Cin, k, k, out.ctypes.data_as(C.c_void_p), C.byref(sc)) == 0
    inv = sc.value                                         # the accumulator multiplier = 1 / scale
    scale = 1.0 / inv
    assert np.log2(scale) == round(np.log2(scale)) and 2.0 ** 14 <= float(w.abs().max()) * scale < 2.0 ** 15
    x = (w.permute(0, 2, 3, 1).reshape(N, K).numpy().astype(np.float32) * np.float32(scale))
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    halves = out.view(np.uint16).reshape(Npad, Kpad // 8, 16)      # [row][group of 8 k][8 hi | 8 lo]
    got_hi = halves[:N, :, :8].reshape(N, -1)[:, :K].view(np.float16)
    got_lo = halves[:N, :, 8:].reshape(N, -1)[:, :K].view(np.float16)
    assert np.array_equal(got_hi.view(np.uint16), hi.view(np.uint16))
    assert np.array_equal(got_lo.view(np.uint16), lo.view(np.uint16))
    rec = (got_hi.astype(np.float64) + got_lo.astype(np.float64)) * inv
    ref = w.permute(0, 2, 3, 1).reshape(N, K).numpy().astype(np.float64)
    assert np.abs(rec - ref).max() <= np.abs(ref).max() * 2.0 ** -21
    assert not out[N:].any() and not halves[:N].reshape(N, -1, 16)[:, (K + 7) // 8:].any()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from unicorn_amd import _lib
    from unicorn_amd.models import Unicorn
    from unicorn_amd.ops import corr_softmax_pv, corr_softmax_pv_batched
    m = Unicorn("unicorn_track_tiny")
    with pytest.raises(_lib.UnicornHipError):
        m.cuda()
    with pytest.raises(_lib.UnicornHipError):
        corr_softmax_pv(torch.zeros(128, 4), torch.zeros(128, 4), torch.zeros(1, 4))
    with pytest.raises(_lib.UnicornHipError):
        corr_softmax_pv_batched(torch.zeros(2, 128, 4), torch.zeros(2, 128, 4), torch.zeros(1, 4))
    with pytest.raises(_lib.UnicornHipError):
        m(imgs=torch.zeros(1, 3, 32, 32), mode="backbone")


def _f16_pairs(blob_u16):
    return blob_u16.view(np.float16).astype(np.float64)


@pytest.mark.parametrize("C_,layout", [(96, 0), (192, 0), (256, 0), (192, 1), (256, 1)])
def test_mlp_pack_layouts_match_numpy_restatement(C_, layout):
    """Host packer of the fused ConvNeXt MLP (csrc/mlp_fused.hip): the weight stream is 2 * NH pieces of 128 C bytes, blob position 2 i =
    W1 slab of hidden block i, 2 i + 1 = W2 slab of hidden block i - 1 (cyclic), every piece in the LDS image of its kernel layout, values
    split into hi + lo f16 halves of w * 2^k.  Restated here from the kernel's fragment addressing and decoded back to the weights."""
    from unicorn_amd import _lib
    lib = _lib.lib()
    rng = np.random.RandomState(C_ + layout)
    w1 = (rng.randn(4 * C_, C_) * 0.05).astype(np.float32)
    w2 = (rng.randn(C_, 4 * C_) * 0.05).astype(np.float32)
    gamma = (rng.rand(C_) + 0.5).astype(np.float32)
    nb = lib.uni_mlp_blob_bytes(C_)
    assert nb == 32 * C_ * C_
    blob = np.zeros(nb // 2, dtype=np.uint16)
    a, b = C.c_float(0), C.c_float(0)
    assert lib.uni_mlp_pack(w1.ctypes.data_as(C.c_void_p), w2.ctypes.data_as(C.c_void_p), gamma.ctypes.data_as(C.c_void_p), C_, layout,
                            blob.ctypes.data_as(C.c_void_p), C.byref(a), C.byref(b)) == 0
    s1, s2 = 1.0 / a.value, 1.0 / b.value
    assert np.log2(s1) == round(np.log2(s1)) and np.log2(s2) == round(np.log2(s2))          # power-of-two scales
    assert 2 ** 14 <= np.abs(w1).max() * s1 < 2 ** 15
    v = _f16_pairs(blob)
    NH, PBh = C_ // 8, 64 * C_                                   # hidden blocks, f16 values per piece
    r1, r2 = np.zeros_like(w1, dtype=np.float64), np.zeros_like(w2, dtype=np.float64)
    for i in range(NH):
        p1 = v[(2 * i) * PBh:(2 * i + 1) * PBh]
        p2 = v[(2 * i + 1) * PBh:(2 * i + 2) * PBh]
        h2 = (i - 1) % NH
        if layout == 0:      # 32-row waves: W1 [slice 16 k][row 32][chunk ^ ((r >> 2) & 3)], W2 [col block 32][row 32][chunk ^ ((r >> 1) & 7)]
            q = p1.reshape(C_ // 16, 32, 4, 8)
            for r in range(32):
                for pc in range(4):
                    c = pc ^ ((r >> 2) & 3)
                    fh, hl = c >> 1, c & 1
                    for s in range(C_ // 16):
                        r1[32 * i + r, 16 * s + 8 * fh:16 * s + 8 * fh + 8] += q[s, r, pc]       # hi + lo
            q = p2.reshape(C_ // 32, 32, 8, 8)
            perm = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]                # k position -> hidden unit inside 16
            for j in range(C_ // 32):
                for r in range(32):
                    for pc in range(8):
                        c = pc ^ ((r >> 1) & 7)
                        s, fh = c >> 2, (c >> 1) & 1
                        for e in range(8):
                            r2[32 * j + r, 32 * h2 + 16 * s + perm[8 * fh + e]] += q[j, r, pc, e]
        else:                # 16-row waves: [..][kg 4][hi, lo][row 16] x 8 values
            q = p1.reshape(2, C_ // 32, 4, 2, 16, 8)
            for ub in range(2):
                for s in range(C_ // 32):
                    for kg in range(4):
                        r1[32 * i + 16 * ub:32 * i + 16 * ub + 16, 32 * s + 8 * kg:32 * s + 8 * kg + 8] = q[ub, s, kg, 0] + q[ub, s, kg, 1]
            q = p2.reshape(C_ // 16, 4, 2, 16, 8)
            for j in range(C_ // 16):
                for kg in range(4):
                    for e in range(8):
                        u = 4 * kg + e if e < 4 else 16 + 4 * kg + e - 4
                        r2[16 * j:16 * j + 16, 32 * h2 + u] = q[j, kg, 0, :, e] + q[j, kg, 1, :, e]
    assert np.abs(r1 / s1 - w1).max() <= np.abs(w1).max() * 2.0 ** -21
    assert np.abs(r2 / s2 - gamma[:, None] * w2).max() <= np.abs(gamma[:, None] * w2).max() * 2.0 ** -21


def test_build_manifest_is_verified_at_load(tmp_path, monkeypatch):
    """The libraries are git-ignored build products that travel to the GPU box as files: csrc/build.sh writes lib/build_manifest.json (sha256
    of every source / header and of both libraries), `_lib.lib()` / `assoc_lib()` hold the library AND the sources next to it to it.  A stale
    library (source changed after the build), an unknown new header and a swapped .so all raise; UNI_SKIP_MANIFEST=1 downgrades to a warning."""
    import json
    import shutil
    import warnings
    from unicorn_amd import _lib as L
    assert L.verify_manifest("libunicorn_hip.so") == "verified" and L.verify_manifest("libunicorn_assoc.so") == "verified"
    L.lib()
    assert L.BUILD_CHECK == "verified"
    man = json.load(open(os.path.join(os.path.dirname(L.LIB_PATH), "build_manifest.json")))
    assert {"common.h", "kernels.h", "engine.h", "gemm_epi.h", "mask_interp.h", "unicorn_hip.h", "unicorn_assoc.h", "engine.hip", "assoc.cpp"} <= set(man["sources"])
    # a copy of the package tree with (a) a touched source, (b) an extra header, (c) a different library
    root = tmp_path / "unicorn_amd"
    shutil.copytree(os.path.dirname(L.__file__), root, ignore=shutil.ignore_patterns("build", "__pycache__"))
    shutil.copytree(os.path.join(os.path.dirname(os.path.dirname(L.__file__)), "include"), tmp_path / "include")
    monkeypatch.setattr(L, "_HERE", str(root))
    assert L.verify_manifest("libunicorn_hip.so") == "verified"
    with open(root / "csrc" / "norm.hip", "a") as f:
        f.write("// edited after the build\n")
    with pytest.raises(L.UnicornHipError, match="norm.hip changed"):
        L.verify_manifest("libunicorn_hip.so")
    monkeypatch.setenv("UNI_SKIP_MANIFEST", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert L.verify_manifest("libunicorn_hip.so") == "skipped" and any("norm.hip" in str(x.message) for x in w)
    monkeypatch.delenv("UNI_SKIP_MANIFEST")
    shutil.copy(os.path.join(os.path.dirname(L.__file__), "csrc", "norm.hip"), root / "csrc" / "norm.hip")
    (root / "csrc" / "brand_new.h").write_text("#pragma once\n")
    with pytest.raises(L.UnicornHipError, match="brand_new.h"):
        L.verify_manifest("libunicorn_hip.so")
    os.remove(root / "csrc" / "brand_new.h")
    with open(root / "lib" / "libunicorn_assoc.so", "ab") as f:
        f.write(b"\0")
    with pytest.raises(L.UnicornHipError, match="not the library"):
        L.verify_manifest("libunicorn_assoc.so")
    os.remove(root / "lib" / "build_manifest.json")
    with pytest.raises(L.UnicornHipError, match="missing"):
        L.verify_manifest("libunicorn_hip.so")


def test_half_is_a_documented_no_op_that_warns_once():
    """tools/track.py --fp16 calls model.half() (mot_evaluator.py:126-128): the operand format is fixed by `precision`, so the call changes
    nothing -- and says so ONCE per process instead of silently costing fp32-equivalent time."""
    import warnings
    from unicorn_amd.models import Unicorn
    Unicorn._half_warned = False
    m = Unicorn("unicorn_track_tiny")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert m.half() is m and m.half() is m
    msgs = [str(x.message) for x in w if "half()" in str(x.message)]
    assert len(msgs) == 1 and "precision='f16x2'" in msgs[0], msgs


def test_flat_weights_file_round_trip(tmp_path):
    """`unicorn_amd.utils.checkpoint.export_flat` / tools/export_weights.py write the deployment artefact a host without Python / torch loads through
    uni_weights_file_cfg + uni_ctx_load_file (include/unicorn_hip.h; the role tools/export_torchscript.py:51-71 has in the reference).  CPU half: the file's header
    is read back by the LIBRARY (no GPU needed for that entry point), every tensor by a byte-level reader; foreign keys are skipped, a missing tensor raises."""
    import ctypes as C
    import struct
    import synth
    import unicorn_oracle as uo
    from unicorn_amd import _lib as L
    from unicorn_amd.utils.checkpoint import export_flat, state_spec
    exp = "unicorn_track_tiny_mask"
    sd = synth.synth_state_dict(uo.CONFIGS[exp])
    path = str(tmp_path / "w.uniw")
    n = export_flat(dict(sd, **{"head.mask_head._iter": torch.zeros(1)}), exp, path, precision="fp32")      # a buffer of a released checkpoint: not in the spec, skipped
    spec = state_spec(exp)
    assert n == len(spec)
    cfg = L.ModelCfg()
    L.check(L.lib().uni_weights_file_cfg(path.encode(), C.byref(cfg)), "uni_weights_file_cfg")
    assert list(cfg.dims) == [96, 192, 384, 768] and list(cfg.depths) == [3, 3, 9, 3] and cfg.mask == 1 and cfg.num_classes == 8 and cfg.precision == 1
    assert cfg.embed_dim == 128 and cfg.up_rate == 4 and cfg.d_rate == 2 and cfg.n_layer_att == 3
    raw = open(path, "rb").read()
    assert raw[:8] == b"UNIW1\0\0\0" and struct.unpack_from("<i", raw, 8 + 60)[0] == n
    off, seen = 8 + 60 + 4, 0
    while off < len(raw):
        (nl,) = struct.unpack_from("<i", raw, off)
        name = raw[off + 4:off + 4 + nl].decode()
        (nd,) = struct.unpack_from("<i", raw, off + 4 + nl)
        shape = struct.unpack_from("<%dq" % nd, raw, off + 8 + nl)
        cnt = int(np.prod(shape)) if nd else 1
        data = np.frombuffer(raw, dtype="<f4", count=cnt, offset=off + 8 + nl + 8 * nd)
        assert tuple(shape) == tuple(spec[name]) and np.array_equal(data, sd[name].numpy().reshape(-1)), name
        off += 8 + nl + 8 * nd + 4 * cnt
        seen += 1
    assert seen == n and off == len(raw)
    bad = dict(sd)
    bad.pop("bottleneck.0.weight")
    with pytest.raises(KeyError, match="bottleneck.0.weight"):
        export_flat(bad, exp, str(tmp_path / "bad.uniw"))
    open(str(tmp_path / "junk.uniw"), "wb").write(b"not a weights file at all")
    assert L.lib().uni_weights_file_cfg(str(tmp_path / "junk.uniw").encode(), C.byref(cfg)) < 0 and b"UNIW1" in L.lib().uni_last_error()
