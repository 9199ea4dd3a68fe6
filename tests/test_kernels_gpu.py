"""Kernel-level parity: every HIP kernel (through the C-ABI) vs a plain fp32 PyTorch/oracle statement of the
same op on the same seeded inputs.  bf16 kernels are compared on bf16-rounded operands so only the
accumulation order differs."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import unicorn_oracle as uo  # noqa: E402


@pytest.fixture(scope="module")
def L():
    from unicorn_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return _lib


def dev(t):
    return t.cuda()


def bf16_round(t):
    return t.to(torch.bfloat16).float()


def as_u16(t):
    """fp32 tensor -> bf16 device tensor (bit pattern, RNE)"""
    return t.to(torch.bfloat16).contiguous()


def pack_weight(L, w):
    N, Cin, KH, KW = w.shape
    K = Cin * KH * KW
    Npad, Kpad = (N + 255) // 256 * 256, (K + 63) // 64 * 64
    out = np.zeros((Npad, Kpad), dtype=np.uint16)
    wc = np.ascontiguousarray(w.float().numpy())
    L.check(L.lib().uni_pack_weight(wc.ctypes.data_as(C.c_void_p), N, Cin, KH, KW, out.ctypes.data_as(C.c_void_p)), "pack")
    return torch.from_numpy(out.view(np.int16)).cuda()


ACTS = {0: lambda x: x, 1: F.relu, 2: F.gelu, 3: F.silu, 4: torch.sigmoid}


@pytest.mark.parametrize("cfg", [0, 22, 12, 21, 11, 122, 42, 24, 44, 444, 445, 224])
@pytest.mark.parametrize("case", [
    # (Hin, Win, Cin, N, KH, stride, pad, act, bias, res, stats_G)
    (20, 24, 96, 384, 1, 1, 0, 2, True, False, 0),        # tiny pwconv1 + GELU, K=96 (padded to 128)
    (20, 24, 384, 96, 1, 1, 0, 0, True, True, 0),         # pwconv2 + residual
    (25, 40, 256, 256, 3, 1, 1, 0, False, False, 16),     # head 3x3 + GN stats, M=1000 (ragged M)
    (26, 34, 192, 192, 3, 2, 1, 0, False, False, 16),     # bu_conv 3x3 stride 2
    (20, 20, 96, 192, 2, 2, 0, 0, True, False, 0),        # downsample 2x2/s2
    (10, 10, 48, 48, 3, 1, 1, 0, False, False, 16),       # tiny CSP bottleneck (cpg=3)
    (13, 17, 256, 5, 1, 1, 0, 4, True, False, 0),         # reg/obj preds N=5 (sigmoid on col>=4 tested below)
    (13, 17, 256, 169, 3, 1, 1, 0, True, False, 0),       # controller N=169
    (16, 16, 64, 256, 3, 1, 1, 1, True, False, 0),        # upsample_layer.1 + ReLU
])
def test_gemm_conv(L, cfg, case):
    Hin, Win, Cin, N, k, stride, pad, act, use_bias, use_res, G = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = bf16_round(torch.randn(1, Cin, Hin, Win, generator=g))
    w = bf16_round(torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    bias = torch.randn(N, generator=g) * 0.1 if use_bias else None
    ref = F.conv2d(x, w, bias, stride=stride, padding=pad)           # fp32 reference on the same operands
    Hout, Wout = ref.shape[2:]
    M = Hout * Wout
    raw = ref.permute(0, 2, 3, 1).reshape(M, N)
    res = torch.randn(M, N, generator=g) if use_res else None
    exp = ACTS[act](raw) + (res if use_res else 0)
    A = as_u16(x.permute(0, 2, 3, 1).reshape(Hin * Win, Cin)).cuda()
    Wp = pack_weight(L, w)
    outF = torch.full((M, N), float("nan"), device="cuda")
    outB = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(64, device="cuda", dtype=torch.float64) if G else None
    bias_d = dev(bias) if use_bias else None
    res_d = dev(res) if use_res else None
    L.check(L.lib().uni_gemm_bf16(L.ptr(A), Cin, L.ptr(Wp), M, N, Hin, Win, Cin, k, k, stride, pad,
                                  L.ptr(bias_d), act, L.ptr(res_d), N,
                                  L.ptr(outF), N, L.ptr(outB), N, L.ptr(stats), (N // G) if G else 0, cfg, L.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    got = outF.cpu()
    assert torch.isfinite(got).all()
    err = (got - exp).abs().max().item()
    assert err < 2e-3 * max(1.0, exp.abs().max().item()), err
    assert (outB.float().cpu() - exp).abs().max().item() < 1e-2 * max(1.0, exp.abs().max().item())
    if G:
        cpg = N // G
        grp = raw.reshape(M, G, cpg)
        s_ref = torch.stack([grp.sum((0, 2)), (grp ** 2).sum((0, 2))], 1).double()
        s_got = stats.cpu()[:2 * G].reshape(G, 2)
        assert torch.allclose(s_got, s_ref, rtol=1e-3, atol=1e-2), (s_got - s_ref).abs().max()


@pytest.mark.parametrize("cfg", [144, 44, 444, 445, 224, 0])
@pytest.mark.parametrize("case", [
    # (M, N, K, act, res, outF, outB): > 256 tiles of 256x128 so persistent blocks walk several tiles
    (70001, 256, 64, 2, False, False, True),      # one K step per tile (drain slices outnumber K steps), GELU, bf16 out
    (35003, 512, 136, 0, True, True, True),       # K tail (136 -> 3 steps), residual + fp32 + bf16 out, ragged M
    (9001, 3072, 320, 2, False, False, True),     # pwconv1-like
    (20000, 768, 1024, 0, True, True, False),     # pwconv2-like, fp32 out only
    (5000, 136, 256, 3, False, True, True),       # N not a multiple of the tile (136 = 128 + 8)
])
def test_gemm_large(L, cfg, case):
    M, N, K, act, use_res, use_F, use_B = case
    g = torch.Generator().manual_seed(M + N + K)
    x = bf16_round(torch.randn(M, K, generator=g)).cuda()
    w = bf16_round(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    res = torch.randn(M, N, generator=g).cuda() if use_res else None
    raw = x @ w.cuda().t() + bias                     # fp32 on the same bf16-rounded operands
    exp = ACTS[act](raw) + (res if use_res else 0)
    A = x.to(torch.bfloat16)
    Wp = pack_weight(L, w.reshape(N, K, 1, 1))
    outF = torch.full((M, N), float("nan"), device="cuda") if use_F else None
    outB = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16) if use_B else None
    L.check(L.lib().uni_gemm_bf16(L.ptr(A), K, L.ptr(Wp), M, N, M, 1, K, 1, 1, 1, 0, L.ptr(bias), act, L.ptr(res), N,
                                  L.ptr(outF), N, L.ptr(outB), N, None, 0, cfg, L.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    scale = max(1.0, exp.abs().max().item())
    if use_F:
        assert torch.isfinite(outF).all()
        assert (outF - exp).abs().max().item() < 2e-3 * scale
    if use_B:
        assert torch.isfinite(outB.float()).all()
        assert (outB.float() - exp).abs().max().item() < 1e-2 * scale


@pytest.mark.parametrize("case", [
    # plain: (M, N, K, act, res, outF, outB);  conv: (Hin, Win, Cin, N, k, stride, pad, act, bias, G, fp32_out)
    ("plain", 70001, 256, 128, 2, False, False, True),     # two 64-k steps per tile (the shortest DMA stream), GELU, bf16 out, ragged M
    ("plain", 35003, 512, 192, 0, True, True, True),       # odd step count (stage parity flips from tile to tile), residual + fp32 + bf16 out
    ("plain", 9001, 3072, 320, 2, False, False, True),     # pwconv1-like
    ("plain", 20000, 768, 1024, 0, True, True, False),     # pwconv2-like, fp32 out only
    ("plain", 5000, 136, 256, 1, False, True, True),       # N not a multiple of the tile, ReLU
    ("conv", 200, 320, 64, 512, 3, 1, 1, 0, False, 16, True),     # 3x3 + GroupNorm sums, one step per tap
    ("conv", 101, 163, 128, 384, 3, 2, 1, 0, True, 16, True),     # 3x3 stride 2, odd map
    ("conv", 120, 160, 64, 256, 2, 2, 0, 0, True, 0, True),       # 2x2 / s2 downsample
    ("conv", 160, 200, 128, 256, 1, 1, 0, 0, False, 16, True),    # 1x1 + GroupNorm sums
    ("conv", 96, 96, 64, 256, 3, 1, 1, 1, True, 0, False),        # 3x3 + ReLU, bf16 output only
])
def test_gemm_bf16_pingpong(L, case):
    """The bf16 instantiations of the ping-pong persistent kernel (gemm_h2q.hip, cfg 188) against fp32 on the same bf16-rounded
    operands; three launches must agree bit for bit (fp64 statistics atomics excepted)."""
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    if case[0] == "plain":
        _, M, N, K, act, use_res, use_F, use_B = case
        x = bf16_round(torch.randn(M, K, generator=g)).cuda()
        w = bf16_round(torch.randn(N, K, generator=g) / K ** 0.5)
        bias = (torch.randn(N, generator=g) * 0.1).cuda()
        res = torch.randn(M, N, generator=g).cuda() if use_res else None
        raw = x @ w.cuda().t() + bias
        A = x.to(torch.bfloat16)
        Wp = pack_weight(L, w.reshape(N, K, 1, 1))
        geo = (M, 1, K, 1, 1, 1, 0)
        G = 0
    else:
        _, Hin, Win, Cin, N, k, stride, pad, act, use_bias, G, use_F = case
        use_B, use_res, res = not use_F, False, None
        x = bf16_round(torch.randn(1, Cin, Hin, Win, generator=g)).cuda()
        w = bf16_round(torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
        bias = (torch.randn(N, generator=g) * 0.1).cuda() if use_bias else None
        ref = F.conv2d(x, w.cuda(), bias, stride=stride, padding=pad)
        M = ref.shape[2] * ref.shape[3]
        raw = ref.permute(0, 2, 3, 1).reshape(M, N)
        A = x.permute(0, 2, 3, 1).reshape(Hin * Win, Cin).contiguous().to(torch.bfloat16)
        Wp = pack_weight(L, w)
        K = Cin
        geo = (Hin, Win, Cin, k, k, stride, pad)
    exp = ACTS[act](raw) + (res if use_res else 0)
    scale = max(1.0, exp.abs().max().item())
    outs = []
    for rep in range(3):
        outF = torch.full((M, N), float("nan"), device="cuda") if use_F else None
        outB = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16) if use_B else None
        stats = torch.zeros(64, device="cuda", dtype=torch.float64) if G else None
        L.check(L.lib().uni_gemm_bf16(L.ptr(A), K, L.ptr(Wp), M, N, *geo, L.ptr(bias), act, L.ptr(res), N,
                                      L.ptr(outF), N, L.ptr(outB), N, L.ptr(stats), (N // G) if G else 0, 188, L.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        if use_F:
            assert torch.isfinite(outF).all() and (outF - exp).abs().max().item() < 2e-3 * scale
        if use_B:
            assert torch.isfinite(outB.float()).all() and (outB.float() - exp).abs().max().item() < 1e-2 * scale
        if G:
            grp = raw.reshape(M, G, N // G).double()
            s_ref = torch.stack([grp.sum((0, 2)), (grp ** 2).sum((0, 2))], 1)
            assert torch.allclose(stats[:2 * G].reshape(G, 2), s_ref, rtol=1e-3, atol=1.0)
        outs.append((outF, outB))
    for o in outs[1:]:
        assert (o[0] is None or torch.equal(o[0], outs[0][0])) and (o[1] is None or torch.equal(o[1], outs[0][1]))


def test_gemm_act_col0(L):
    g = torch.Generator().manual_seed(5)
    M, K, N = 300, 256, 5
    x = bf16_round(torch.randn(M, K, generator=g))
    w = bf16_round(torch.randn(N, K, 1, 1, generator=g) / 16)
    b = torch.randn(N, generator=g)
    raw = x @ w.reshape(N, K).t() + b
    exp = raw.clone()
    exp[:, 4:] = torch.sigmoid(exp[:, 4:])
    out = torch.zeros((M, 6), device="cuda")           # ld 6 like the (A, 5+nc) head buffer
    lib = L.lib()
    # act_col0 is only reachable through the engine; emulate with two calls is not possible -> call engine-level check in
    # test_model_gpu; here check ragged ldf writes leave the 6th column untouched
    xd, wd, bd = as_u16(x).cuda(), pack_weight(L, w), b.cuda()
    L.check(lib.uni_gemm_bf16(L.ptr(xd), K, L.ptr(wd), M, N, M, 1, K, 1, 1, 1, 0, L.ptr(bd), 0,
                              None, 0, L.ptr(out), 6, None, 0, None, 0, 0, L.stream_ptr()), "gemm")
    torch.cuda.synchronize()
    assert (out[:, :5].cpu() - raw).abs().max() < 2e-3 * raw.abs().max()
    assert (out[:, 5] == 0).all()


@pytest.mark.parametrize("C_", [96, 192, 256, 384, 768, 1536])
def test_layernorm(L, C_):
    g = torch.Generator().manual_seed(C_)
    M = 777
    x = torch.randn(M, C_, generator=g) * 3 + 1.5
    ga, be = torch.randn(C_, generator=g), torch.randn(C_, generator=g)
    exp = F.layer_norm(x, (C_,), ga, be, 1e-6)
    outF = torch.empty((M, C_), device="cuda")
    outB = torch.empty((M, C_), device="cuda", dtype=torch.bfloat16)
    xd, gd, bd = x.cuda(), ga.cuda(), be.cuda()
    L.check(L.lib().uni_layernorm(L.ptr(xd), C_, L.ptr(gd), L.ptr(bd), 1e-6, M, C_, L.ptr(outF), L.ptr(outB),
                                  L.stream_ptr()), "ln")
    torch.cuda.synchronize()
    assert (outF.cpu() - exp).abs().max() < 2e-5 * exp.abs().max()
    assert (outB.float().cpu() - exp).abs().max() < 8e-3 * exp.abs().max()


@pytest.mark.parametrize("shape", [(96, 20, 24), (192, 13, 10), (256, 25, 40), (384, 9, 16), (768, 10, 10), (1536, 5, 8),
                                   (192, 101, 163), (192, 151, 163), (768, 49, 83),     # 8-px and 2-row variants
                                   (768, 127, 163), (256, 207, 323), (192, 261, 317), (384, 255, 163)])   # persistent LDS-weight variant (in-wave / LDS reduction)
def test_dwconv7_ln(L, shape):
    C_, H, W = shape
    g = torch.Generator().manual_seed(C_ + H)
    x = torch.randn(1, C_, H, W, generator=g)
    w = torch.randn(C_, 1, 7, 7, generator=g) / 7
    b, ga, be = torch.randn(C_, generator=g) * 0.1, 1 + 0.1 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    y = F.conv2d(x, w, b, padding=3, groups=C_).permute(0, 2, 3, 1)
    exp = F.layer_norm(y, (C_,), ga, be, 1e-6).reshape(H * W, C_)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    wt = w.reshape(C_, 49).t().contiguous().cuda()
    out = torch.empty((H * W, C_), device="cuda", dtype=torch.bfloat16)
    bd, gd, bed = b.cuda(), ga.cuda(), be.cuda()
    L.check(L.lib().uni_dwconv7_ln(L.ptr(xn), L.ptr(wt), L.ptr(bd), L.ptr(gd), L.ptr(bed), 1e-6, H, W, C_,
                                   L.ptr(out), L.stream_ptr()), "dwln")
    torch.cuda.synchronize()
    err = (out.float().cpu() - exp).abs().max().item()
    assert err < 8e-3 * max(1.0, exp.abs().max().item()), err      # bf16 output rounding only


def _dwln_ref(x, w, b, ga, be):
    C_ = x.shape[1]
    y = F.conv2d(x, w, b, padding=3, groups=C_).permute(0, 2, 3, 1)
    return F.layer_norm(y, (C_,), ga, be, 1e-6).reshape(-1, C_)


def _dwln_decode(out, fmt, M, C_):
    if fmt == 1:
        return out.float().cpu()
    if fmt == 0:
        return out.float().cpu()
    g = out.view(torch.float16).reshape(M, C_ // 8, 16).float().cpu()      # f16x2: [8 hi][8 lo] per 8 channels
    return (g[:, :, :8] + g[:, :, 8:]).reshape(M, C_)


@pytest.mark.parametrize("fmt", [2, 1, 0])
@pytest.mark.parametrize("shape", [(768, 6, 49, 83), (192, 2, 101, 163), (256, 3, 57, 90), (384, 4, 50, 81), (512, 5, 33, 70), (96, 3, 20, 24),
                                   (768, 1, 10, 10), (768, 1, 50, 80), (768, 2, 31, 45), (768, 1, 45, 83), (768, 1, 49, 83),      # row-split one-frame kernel (dwconv7_lns: <= 256 strips), odd H / ragged W / two samples; 275 strips: the fallback
                                   (192, 3, 150, 323), (384, 3, 127, 163), (192, 2, 201, 320)])      # packed-lane strip groups (ragged last group, full rows)
def test_dwconv7_ln_batched_all_formats(L, shape, fmt):
    """uni_dwconv7_ln_ex = the call the engine makes per ConvNeXt block: B stacked maps, every operand format (bf16 / fp32 / f16x2 rows),
    widths that are no multiple of the 8-px strips, odd heights (2- and 4-row kernels), against torch's depthwise conv + LayerNorm;
    samples must not leak into each other (zero padding between stacked maps)."""
    C_, B, H, W = shape
    g = torch.Generator().manual_seed(C_ + H + B)
    x = torch.randn(B, C_, H, W, generator=g)
    w = torch.randn(C_, 1, 7, 7, generator=g) / 7
    b, ga, be = torch.randn(C_, generator=g) * 0.1, 1 + 0.1 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    exp = _dwln_ref(x, w, b, ga, be)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    wt = w.reshape(C_, 49).t().contiguous().cuda()
    bd, gd, bed = b.cuda(), ga.cuda(), be.cuda()
    M = B * H * W
    out = torch.full((M, C_), 7.0, device="cuda", dtype=torch.bfloat16 if fmt == 0 else torch.float32)
    L.check(L.lib().uni_dwconv7_ln_ex(L.ptr(xn), L.ptr(wt), L.ptr(bd), L.ptr(gd), L.ptr(bed), 1e-6, B, H, W, C_, L.ptr(out), fmt, L.stream_ptr()), "dwln_ex")
    torch.cuda.synchronize()
    o = _dwln_decode(out, fmt, M, C_)
    tol = 1.6e-2 if fmt == 0 else 2e-5            # bf16: half an ulp of the largest output
    assert (o - exp).abs().max().item() < tol * max(1.0, exp.abs().max().item())


@pytest.mark.parametrize("C_,G,act", [(256, 16, 3), (192, 16, 3), (48, 16, 3), (256, 32, 0), (128, 16, 1)])
def test_groupnorm_act(L, C_, G, act):
    g = torch.Generator().manual_seed(C_ + G)
    M = 1000
    x = torch.randn(M, C_, generator=g) * 2 + 0.5
    ga, be = torch.randn(C_, generator=g), torch.randn(C_, generator=g)
    eps = 1e-3
    xn = x.t().reshape(1, C_, M, 1)
    exp = ACTS[act](F.group_norm(xn, G, ga, be, eps)).reshape(C_, M).t()
    grp = x.reshape(M, G, C_ // G).double()
    stats = torch.stack([grp.sum((0, 2)), (grp ** 2).sum((0, 2))], 1).reshape(-1).cuda()
    outF = torch.empty((M, C_), device="cuda")
    xd, gd, bd = x.cuda(), ga.cuda(), be.cuda()
    L.check(L.lib().uni_groupnorm_act(L.ptr(xd), L.ptr(stats), L.ptr(gd), L.ptr(bd), eps, M, C_, G, act,
                                      L.ptr(outF), None, L.stream_ptr()), "gn")
    torch.cuda.synchronize()
    assert (outF.cpu() - exp).abs().max() < 1e-4 * max(1.0, exp.abs().max().item())


@pytest.mark.parametrize("W", [96, 100, 1280])     # W % 16 == 0: 4-pixel kernel, otherwise the 1-pixel fallback
@pytest.mark.parametrize("C_", [96, 192])
def test_stem(L, C_, W):
    g = torch.Generator().manual_seed(C_)
    H = 64
    img = torch.rand(1, 3, H, W, generator=g) * 255
    w = torch.randn(C_, 3, 4, 4, generator=g) / 48 ** 0.5
    b, ga, be = torch.randn(C_, generator=g) * 0.1, 1 + 0.1 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    exp = uo.ln_channels_first(F.conv2d(img, w, b, stride=4), ga, be).permute(0, 2, 3, 1).reshape(-1, C_)
    wt = w.reshape(C_, 48).t().contiguous().cuda()
    out = torch.empty((H // 4 * W // 4, C_), device="cuda")
    imd, bd, gd, bed = img.cuda(), b.cuda(), ga.cuda(), be.cuda()
    L.check(L.lib().uni_stem(L.ptr(imd), H, W, L.ptr(wt), L.ptr(bd), L.ptr(gd), L.ptr(bed), C_, L.ptr(out),
                             L.stream_ptr()), "stem")
    torch.cuda.synchronize()
    assert (out.cpu() - exp).abs().max() < 2e-4 * max(1.0, exp.abs().max().item())


def test_msda_known_answer(L, golden_dir):
    """The reference's own test shapes/seed (unicorn/models/ops/test.py:24-50) + out-of-range samples."""
    from unicorn_amd.ops import msda_forward
    g = np.load(os.path.join(golden_dir, "msda_known_answer.npz"))
    for sfx, tol in (("", 1e-7), ("2", 1e-5)):
        shapes = torch.from_numpy(g["shapes" + sfx])
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        out = msda_forward(torch.from_numpy(g["value" + sfx]).cuda(), shapes, lsi, torch.from_numpy(g["loc" + sfx]).cuda(),
                           torch.from_numpy(g["attn" + sfx]).cuda(), 64)
        ref = torch.from_numpy(g["out" + sfx])
        assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=tol), (out.cpu() - ref).abs().max()   # test.py:47 uses rtol 1e-2


def test_msda_random_vs_oracle(L):
    from unicorn_amd.ops import msda_forward
    g = torch.Generator().manual_seed(11)
    shapes = [(50, 80), (50, 80)]
    N, M, D, Lq, P = 1, 8, 32, 700, 4
    S = sum(h * w for h, w in shapes)
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, 2, P, 2, generator=g) * 1.2 - 0.1
    attn = torch.softmax(torch.randn(N, Lq, M, 8, generator=g), -1).view(N, Lq, M, 2, P)
    ref = uo.msda_core(value, shapes, loc, attn)
    shp = torch.tensor(shapes)
    lsi = torch.tensor([0, 4000])
    out = msda_forward(value.cuda(), shp, lsi, loc.cuda(), attn.cuda())
    assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=1e-5)
    # empty query set (edge case): Lq = 0
    out0 = msda_forward(value.cuda(), shp, lsi, loc[:, :0].cuda(), attn[:, :0].cuda())
    assert out0.shape == (1, 0, 256)


@pytest.mark.parametrize("B,h,w", [(1, 50, 80), (2, 10, 13), (1, 7, 5)])
def test_msda_wave_kernel_tokens(L, B, h, w):
    """The engine's sampler (msda_wave_kernel: one wave per (token, head)) at kernel level against the oracle's statement of
    MSDeformAttn.forward (ms_deform_attn.py:98-105: softmax over the 8 logits, loc = ref + off / (W, H)) + msda_core, including
    offsets that leave the map (zero padding per corner, samples skipped outside (-1, W) x (-1, H))."""
    g = torch.Generator().manual_seed(B * 100 + h)
    hw = h * w
    Lq = 2 * hw
    value = torch.randn(B, Lq, 256, generator=g)
    off = torch.randn(B, Lq, 8, 2, 4, 2, generator=g) * 6.0          # pixels: many samples land outside or on the border
    off[:, :5] *= 30.0
    logits = torch.randn(B, Lq, 8, 8, generator=g) * 2.0
    offaw = torch.cat([off.reshape(B * Lq, 128), logits.reshape(B * Lq, 64)], 1).contiguous()
    out = torch.empty(B * Lq, 256, device="cuda")
    vd, od = value.cuda(), offaw.cuda()                              # (kept alive across the asynchronous launch)
    L.check(L.lib().uni_msda_tokens(L.ptr(vd), L.ptr(od), 192, B, h, w, L.ptr(out), L.stream_ptr()), "uni_msda_tokens")
    torch.cuda.synchronize()
    # reference points (deformable_transformer.py:141-153): ((j + 0.5) / W, (i + 0.5) / H), the same for both levels / frames
    ii, jj = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    ref = torch.stack([(jj.reshape(-1) + 0.5) / w, (ii.reshape(-1) + 0.5) / h], -1).repeat(2, 1)       # (Lq, 2)
    norm = torch.tensor([w, h], dtype=torch.float32)
    loc = ref[None, :, None, None, None, :] + off / norm
    attn = torch.softmax(logits, -1).view(B, Lq, 8, 2, 4)
    want = uo.msda_core(value.view(B, Lq, 8, 32), [(h, w), (h, w)], loc, attn).reshape(B * Lq, 256)
    assert torch.allclose(out.cpu(), want, rtol=1e-4, atol=2e-5), (out.cpu() - want).abs().max()


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("R,Q,K", [(1600, 1600, 1), (1000, 1300, 3), (4000, 2000, 5), (333, 257, 9), (1500, 1100, 16), (700, 900, 21)])
def test_corr_softmax_pv(L, R, Q, K, prec):
    from unicorn_amd.ops import corr_softmax_pv
    g = torch.Generator().manual_seed(R + Q)
    er = torch.randn(128, R, generator=g) * 0.6
    ec = torch.randn(128, Q, generator=g) * 0.6
    v = torch.rand(K, R, generator=g)
    ref = uo.correlation_propagate(er, ec, v)
    out = corr_softmax_pv(er.cuda(), ec.cuda(), v.cuda(), precision=prec)
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("case", [
    # (B, R, Q, K, values per frame, precision)
    (3, 1600, 1600, 1, False, 2),        # the time-batched SOT step: one shared label map
    (4, 1000, 1300, 3, True, 2),         # value rows per frame, ragged tiles
    (2, 4000, 16000, 1, False, 2),       # enough blocks per frame that the reference axis is still split (partial results + batched merge)
    (5, 700, 900, 16, True, 3),          # 16 value rows, the driver's fp16 class
    (2, 1500, 1100, 21, False, 2),       # more than 16 value rows: frame by frame
    (3, 333, 257, 2, True, 0),           # exact-fp32 kernel: frame by frame
])
def test_corr_softmax_pv_batched(L, case):
    """uni_corr_softmax_pv_batched (B frames in one launch) against the oracle per frame and against the per-frame entry point"""
    from unicorn_amd.ops import corr_softmax_pv, corr_softmax_pv_batched
    B, R, Q, K, vpf, prec = case
    g = torch.Generator().manual_seed(B + R + Q)
    er = torch.randn(B, 128, R, generator=g) * 0.6
    ec = torch.randn(B, 128, Q, generator=g) * 0.6
    v = torch.rand((B, K, R) if vpf else (K, R), generator=g)
    out = corr_softmax_pv_batched(er.cuda(), ec.cuda(), v.cuda(), precision=prec, values_per_frame=vpf)
    assert out.shape == (B, K, Q)
    for b in range(B):
        vb = v[b] if vpf else v
        one = corr_softmax_pv(er[b].cuda(), ec[b].cuda(), vb.cuda(), precision=prec)
        assert (out[b] - one).abs().max().item() < 5e-6          # same kernel, another split of the reference axis (fp32 merge order)
        if prec != 3 and Q <= 2000:
            ref = uo.correlation_propagate(er[b], ec[b], vb)
            assert (out[b].cpu() - ref).abs().max().item() < 2e-5
    with pytest.raises(ValueError):
        corr_softmax_pv_batched(er.cuda(), ec.cuda(), torch.rand(K + 1, 2, R).cuda(), precision=prec, values_per_frame=False)


def test_corr_split_matches_fp32_mfma(L):
    """bf16x3 split (precision 1) against the exact fp32 MFMA kernel (precision 0) at 800x1280 scale with large logits
    (|logit| up to ~60, where softmax amplifies any contraction error), and against an fp64 evaluation"""
    from unicorn_amd.ops import corr_softmax_pv
    g = torch.Generator().manual_seed(11)
    R = Q = 16000
    er = (torch.randn(128, R, generator=g) * 1.3).cuda()
    ec = (torch.randn(128, Q, generator=g) * 1.3).cuda()
    v = torch.rand(3, R, generator=g).cuda()
    o0 = corr_softmax_pv(er, ec, v, precision=0)
    o1 = corr_softmax_pv(er, ec, v, precision=1)
    o2 = corr_softmax_pv(er, ec, v, precision=2)     # f16x2 split: 3 products per slice
    qs = torch.arange(0, Q, 97, device="cuda")
    ref = (v.double() @ torch.softmax(er.double().t() @ ec[:, qs].double(), 0)).float()
    e0 = (o0[:, qs] - ref).abs().max().item()
    e1 = (o1[:, qs] - ref).abs().max().item()
    e2 = (o2[:, qs] - ref).abs().max().item()
    assert (o0 - o1).abs().max().item() < 2e-5 and (o0 - o2).abs().max().item() < 2e-5
    assert e1 < max(3 * e0, 2e-6), (e0, e1)     # same error class as the fp32 MFMA path
    assert e2 < max(4 * e0, 3e-6), (e0, e2)     # 22 operand bits instead of 24


def test_corr_fp16_single_pass_mode(L):
    """precision 3 = the reference driver's fp16 arithmetic class (unicorn_sot.py:95-100) against the oracle's emulation of those
    casts; it must sit closer to that than to the fp32 definition when the logits are large enough for the casts to matter."""
    from unicorn_amd.ops import corr_softmax_pv
    g = torch.Generator().manual_seed(21)
    R, Q = 4000, 1500
    er = torch.randn(128, R, generator=g) * 0.9
    ec = torch.randn(128, Q, generator=g) * 0.9
    v = torch.rand(3, R, generator=g)
    ref16 = uo.correlation_propagate(er, ec, v, half=True)
    ref32 = uo.correlation_propagate(er, ec, v)
    out = corr_softmax_pv(er.cuda(), ec.cuda(), v.cuda(), precision=3).cpu()
    e16, e32 = (out - ref16).abs(), (out - ref32).abs()
    assert e16.max() < 1e-3 and e16.mean() < 1.5e-4, (e16.max(), e16.mean())    # rest: the un-reproduced trans.half() + summation order (CPU emulation: 3e-4 / 6e-5)
    assert e16.mean() < 0.5 * e32.mean(), (e16.mean(), e32.mean())


def test_corr_spiked_rescale(L):
    """force the online-softmax rescale branch: one reference row dominates late in the stream"""
    from unicorn_amd.ops import corr_softmax_pv
    g = torch.Generator().manual_seed(3)
    R, Q = 2048, 256
    er = torch.randn(128, R, generator=g) * 0.3
    ec = torch.randn(128, Q, generator=g) * 0.3
    er[:, 1900] = ec[:, 7] * 40          # huge logit for query 7 at a late tile
    er[:, 5] = ec[:, 100] * 40           # and at the first tile for query 100
    v = torch.rand(2, R, generator=g)
    ref = uo.correlation_propagate(er, ec, v)
    for prec in (0, 1, 2):
        out = corr_softmax_pv(er.cuda(), ec.cuda(), v.cuda(), precision=prec).cpu()
        assert (out - ref).abs().max() < 2e-5
        assert abs(out[0, 7] - v[0, 1900]) < 1e-4 and abs(out[1, 100] - v[1, 5]) < 1e-4


def test_prior_pyramid_label_map(L):
    from unicorn_amd.ops import prior_pyramid, label_map_s8
    g = torch.Generator().manual_seed(9)
    c = torch.rand(1, 3, 100, 160, generator=g)
    ref = uo.prior_pyramid(c)
    got = prior_pyramid(c.cuda())
    for a, b in zip(got, ref):
        assert torch.allclose(a.cpu(), b, atol=1e-6)
    for box in ([320.0, 200.0, 640.0, 400.0], [3.4, 7.5, 1279.6, 700.5], [-20.0, -5.0, 50.5, 2000.0], [100.5, 100.5, 100.5, 300.0]):
        ref = uo.label_map_s8(torch.tensor(box), 800, 1280)
        got = label_map_s8(box, 800, 1280, "cuda")
        assert torch.equal(got.cpu(), ref), box
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "labelmap_ref.npz"))      # the reference's own get_label_map
    for tag in ("a", "b"):
        H, W = (int(v) for v in gold["hw_" + tag])
        for b, ref in zip(gold["boxes"], gold["lbs_" + tag]):
            assert np.array_equal(label_map_s8([float(v) for v in b], H, W, "cuda")[0].cpu().numpy(), ref), (tag, b)


def test_sample_embeddings(L):
    from unicorn_amd.ops import sample_embeddings
    g = torch.Generator().manual_seed(4)
    emb = torch.randn(1, 128, 40, 64, generator=g)
    boxes = torch.rand(50, 7, generator=g) * torch.tensor([512, 320, 512, 320, 1, 1, 1.0])
    boxes[0, :4] = torch.tensor([-30.0, -10.0, 5.0, 6.0])      # centre clipped by border padding
    boxes[1, :4] = torch.tensor([500.0, 300.0, 530.0, 345.0])
    ref = uo.sample_instance_embeddings(emb, boxes[:, :4])
    got = sample_embeddings(emb.cuda(), boxes.cuda())
    assert torch.allclose(got.cpu(), ref, atol=1e-5)
    assert sample_embeddings(emb.cuda(), boxes[:0].cuda()).shape == (0, 128)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_embed_ref.npz"))
    for tag in ("a", "b"):      # rows produced by exec-ing the reference's own lines (tests/golden/make_golden_sample.py)
        got = sample_embeddings(torch.from_numpy(gold["embed_" + tag]).cuda(), torch.from_numpy(gold["boxes_" + tag]).cuda())
        assert torch.allclose(got.cpu(), torch.from_numpy(gold["feats_" + tag]), atol=1e-5), tag


def test_condinst_masks(L):
    from unicorn_amd.ops import condinst_masks
    g = torch.Generator().manual_seed(8)
    H8, W8, n = 20, 28, 5
    cfg = uo.CONFIGS["unicorn_track_tiny_mask"]
    mf = torch.randn(1, 8, H8, W8, generator=g)
    um = torch.randn(1, 144, H8, W8, generator=g)
    params = torch.randn(n, 169, generator=g) * 0.5
    loc = torch.rand(n, 2, generator=g) * torch.tensor([W8 * 8.0, H8 * 8.0])
    lvl = torch.tensor([0, 1, 2, 0, 1])
    ref4 = uo.dynamic_mask_head(cfg, mf, params, loc, lvl, um)
    ref = uo.aligned_bilinear(ref4, 2)
    got4 = condinst_masks(mf.cuda(), um.cuda(), params.cuda(), loc.cuda(), lvl, 4, 1)
    got = condinst_masks(mf.cuda(), um.cuda(), params.cuda(), loc.cuda(), lvl, 4, 2)
    assert torch.allclose(got4.cpu(), ref4, atol=2e-5)
    assert torch.allclose(got.cpu(), ref, atol=2e-5)
    assert condinst_masks(mf.cuda(), um.cuda(), params[:0].cuda(), loc[:0].cuda(), lvl[:0], 4, 2).shape == (0, 1, 160, 224)


@pytest.mark.parametrize("H8,W8,r,H,W", [(40, 64, 0.6667, 480, 768),        # the 1080p geometry (r = 800 / 1200) at a small size
                                         (40, 64, 1.0, 320, 512),            # identity resize
                                         (25, 40, 0.8333333, 250, 390),      # ragged: output tile and window edges, H / W not multiples of the tile
                                         (40, 64, 1.4988, 214, 342),         # image smaller than the network input; output one pixel short of img (853-style)
                                         (40, 64, 2.7, 118, 189),            # strong down-scaling: the tile's window exceeds the LDS -> sample-per-tap path
                                         (100, 160, 0.66666667, 1080, 1920)])  # the bench geometry
def test_condinst_resized_fused_equals_two_pass(L, H8, W8, r, H, W):
    """uni_condinst_masks_u8 (CondInst convex upsample -> aligned bilinear x d_rate -> 1/r bilinear -> `> thr`, chained through LDS) must be
    BIT-IDENTICAL to uni_condinst_masks followed by uni_mask_resize (the (n, Hn, Wn) fp32 maps in HBM): bytes and fp32 probabilities."""
    from unicorn_amd.ops import condinst_masks, condinst_masks_resized, mask_resize
    g = torch.Generator().manual_seed(H8 * 7 + W)
    n = 7
    mf = torch.randn(1, 8, H8, W8, generator=g).cuda()
    um = torch.randn(1, 144, H8, W8, generator=g).cuda()
    params = (torch.randn(n, 169, generator=g) * 0.5).cuda()
    loc = (torch.rand(n, 2, generator=g) * torch.tensor([W8 * 8.0, H8 * 8.0])).cuda()
    lvl = torch.tensor([0, 1, 2, 0, 1, 3, 4])
    for d_rate in (2, 1):
        full = condinst_masks(mf, um, params, loc, lvl, 4, d_rate)[:, 0]
        for thr in (0.3, 0.5, None):
            ref = mask_resize(full, r, H, W, thr=thr)
            got = condinst_masks_resized(mf, um, params, loc, lvl, 4, d_rate, r, H, W, thr=thr)
            assert got.dtype == ref.dtype and got.shape == ref.shape
            assert torch.equal(got, ref), (d_rate, thr, int((got != ref).sum()))
        assert 0.02 < float(mask_resize(full, r, H, W, thr=0.5).float().mean()) < 0.98          # the threshold actually cuts through the maps
    assert condinst_masks_resized(mf, um, params[:0], loc[:0], lvl[:0], 4, 2, r, H, W, thr=0.3).shape == (0, H, W)


from planted import planted_pred as _planted_pred  # noqa: E402


@pytest.mark.parametrize("A,nc,agnostic", [(2100, 1, False), (21000, 1, True), (21000, 8, False), (21000, 8, True), (333, 3, False)])
def test_postprocess_device(L, A, nc, agnostic):
    """uni_postprocess == the oracle's postprocess (boxes.py:33-77 + torchvision nms semantics): same rows, same order, same
    anchor indices, and the in-place corner conversion of the input"""
    from unicorn_amd.ops import postprocess_image
    pred = _planted_pred(A, nc, seed=A + nc)
    ref_in = pred.clone()
    ref_det, ref_idx = uo.postprocess(ref_in, nc, 0.2, 0.45, class_agnostic=agnostic, return_index=True)[0]
    d = pred.clone().cuda()
    det, idx = postprocess_image(d[0], nc, 0.2, 0.45, class_agnostic=agnostic)
    assert ref_det is not None and det is not None and 10 < det.shape[0] < A
    assert torch.equal(idx.cpu(), ref_idx)
    assert torch.equal(det.cpu(), ref_det)
    assert torch.equal(d.cpu()[..., :4], ref_in[..., :4])        # corners written back in place


def test_postprocess_device_empty_and_all(L):
    from unicorn_amd.ops import postprocess_image
    pred = _planted_pred(500, 2, seed=1)
    det, idx = postprocess_image(pred.clone().cuda()[0], 2, 1.5, 0.45)           # nothing passes the confidence cut
    assert det is None and idx is None
    ref_det, ref_idx = uo.postprocess(pred.clone(), 2, 0.0, 0.45, return_index=True)[0]     # every anchor is a candidate
    det, idx = postprocess_image(pred.clone().cuda()[0], 2, 0.0, 0.45)
    assert torch.equal(idx.cpu(), ref_idx) and torch.equal(det.cpu(), ref_det)


def test_nms_wrappers(L):
    from unicorn_amd.utils.boxes import nms, batched_nms
    g = torch.Generator().manual_seed(4)
    n = 700
    xy = torch.rand(n, 2, generator=g) * 300
    boxes = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 80 + 5], 1)
    scores = torch.rand(n, generator=g)
    cls = torch.randint(0, 4, (n,), generator=g)
    assert torch.equal(nms(boxes.cuda(), scores.cuda(), 0.5).cpu(), uo.nms(boxes, scores, 0.5))
    assert torch.equal(batched_nms(boxes.cuda(), scores.cuda(), cls.cuda(), 0.5).cpu(), uo.batched_nms(boxes, scores, cls, 0.5))
    assert nms(boxes[:0].cuda(), scores[:0].cuda(), 0.5).numel() == 0


@pytest.mark.parametrize("shape,size,swap", [((1080, 1920), (800, 1280), True), ((480, 640), (800, 1280), True),
                                             ((375, 1242), (800, 1280), False), ((97, 61), (320, 320), True),
                                             ((800, 1280), (800, 1280), False), ((2160, 3840), (800, 1280), True)])
def test_letterbox_device(L, shape, size, swap):
    """uni_letterbox == the oracle's restatement of PreprocessorX.process / preproc (cv2 8-bit INTER_LINEAR), bit exact"""
    import letterbox_oracle as lo
    from unicorn_amd.ops import letterbox
    g = np.random.default_rng(shape[0] + shape[1])
    img = g.integers(0, 256, shape + (3,), dtype=np.uint8)
    ref, r_ref = lo.letterbox(img, size, swap)
    out, r = letterbox(img, size, swap_rb=swap)
    assert r == r_ref and out.shape == (1, 3) + size
    assert np.array_equal(out[0].cpu().numpy(), ref)


def test_decode_outputs_device(L):
    """uni_decode_outputs == UnicornHead.decode_outputs (unicorn_head.py:467-482) as restated by the oracle, in place"""
    g = torch.Generator().manual_seed(2)
    H, W, nch, B = 320, 352, 6, 2
    levels = [torch.randn(B, nch, H // s_, W // s_, generator=g) for s_ in (8, 16, 32)]
    exp, _ = uo.decode_outputs([t.clone() for t in levels])
    raw = torch.cat([t.flatten(2) for t in levels], 2).permute(0, 2, 1).contiguous()
    d = raw.clone().cuda()
    L.check(L.lib().uni_decode_outputs(L.ptr(d), B, H, W, nch, L.stream_ptr()), "decode")
    torch.cuda.synchronize()
    assert torch.allclose(d.cpu(), exp, rtol=1e-6, atol=1e-5)


# ------------------------------------------------------------------------------------------------
# "f16x2" operand format (precision 2): split-f16 buffers, fp32-equivalent GEMM (gemm_h2.hip)
# ------------------------------------------------------------------------------------------------
def h2_decode(buf, M, C):
    """(M, C) FMT_H2 device buffer (int32 storage, 4 B per element) -> (hi + lo) fp32, and the two halves"""
    h = buf.view(torch.float16).reshape(M, C // 8, 2, 8).float()
    return (h[:, :, 0] + h[:, :, 1]).reshape(M, C), h[:, :, 0].reshape(M, C), h[:, :, 1].reshape(M, C)


def cast_h2(L, x):
    M, C_ = x.shape
    out = torch.zeros((M, C_), device="cuda", dtype=torch.int32)
    L.check(L.lib().uni_cast_h2(L.ptr(x), C_, L.ptr(out), C_, M, C_, L.stream_ptr()), "cast_h2")
    return out


def pack_weight_h2(L, w):
    N, Cin, KH, KW = w.shape
    K = Cin * KH * KW
    Npad, Kpad = (N + 255) // 256 * 256, (K + 63) // 64 * 64
    out = np.zeros((Npad, Kpad), dtype=np.uint32)
    wc = np.ascontiguousarray(w.float().numpy())
    sc = C.c_float(0)
    L.check(L.lib().uni_pack_weight_h2(wc.ctypes.data_as(C.c_void_p), N, Cin, KH, KW, out.ctypes.data_as(C.c_void_p), C.byref(sc)), "pack_h2")
    return torch.from_numpy(out.view(np.int32)).cuda(), sc.value


def test_h2_cast_and_pack_formats(L):
    """hi = f16(x), lo = f16(x - hi): 22 significand bits; tiny values keep an absolute error <= 2^-25 (f16 subnormals are
    kept, not flushed); |x| > 65504 SATURATES at +-65504 (hi = +-65504, lo = 0: never inf / NaN).  The host packer applies a
    power-of-two scale and the same split."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(257, 64, generator=g)
    x[0, :8] = torch.tensor([0.0, 1e-7, -3e-6, 65504.0, 1e5, -1.2e9, 6.1e-5, 1.0])   # beyond 65504: saturation
    x[1] *= 1e-3
    x[2] *= 300.0
    dec, hi, lo = h2_decode(cast_h2(L, x.cuda()), 257, 64)
    dec, hi = dec.cpu(), hi.cpu()
    assert torch.isfinite(dec).all() and torch.isfinite(lo).all()
    assert torch.equal(hi[1:], x[1:].half().float())
    xs = x.clamp(-65504.0, 65504.0)
    assert dec[0, 4] == 65504.0 and dec[0, 5] == -65504.0 and lo.cpu()[0, 4] == 0 and lo.cpu()[0, 5] == 0
    err = (dec - xs).abs()
    assert (err <= xs.abs() * 2.0 ** -21 + 2.0 ** -25).all(), err.max()
    w = torch.randn(40, 16, 3, 3, generator=g) * 0.02
    Wp, wscale = pack_weight_h2(L, w)
    K = 16 * 9
    wd, _, _ = h2_decode(Wp[:40], 40, Wp.shape[1])
    wd = wd[:, :K].cpu() * wscale
    ref = w.permute(0, 2, 3, 1).reshape(40, K)            # (ky, kx, c) order
    assert np.log2(1.0 / wscale) == round(np.log2(1.0 / wscale))
    assert (wd - ref).abs().max() <= ref.abs().max() * 2.0 ** -21


@pytest.mark.parametrize("cfg", [0, 44, 22, 12, 21, 11])
@pytest.mark.parametrize("case", [
    # (Hin, Win, Cin, N, KH, stride, pad, act, bias, res, stats_G)
    (20, 24, 96, 384, 1, 1, 0, 2, True, False, 0),        # pwconv1 + GELU, K = 96 (3 steps of 32)
    (20, 24, 384, 96, 1, 1, 0, 0, True, True, 0),         # pwconv2 + residual
    (25, 40, 256, 256, 3, 1, 1, 0, False, False, 16),     # head 3x3 + GN stats, ragged M
    (26, 34, 192, 192, 3, 2, 1, 0, False, False, 16),     # 3x3 stride 2
    (20, 20, 96, 192, 2, 2, 0, 0, True, False, 0),        # downsample 2x2/s2
    (10, 10, 48, 48, 3, 1, 1, 0, False, False, 16),       # cpg = 3
    (13, 17, 256, 5, 1, 1, 0, 4, True, False, 0),         # reg/obj preds N = 5 (fp32 out only)
    (13, 17, 256, 169, 3, 1, 1, 0, True, False, 0),       # controller N = 169
    (16, 16, 64, 256, 3, 1, 1, 1, True, False, 0),        # upsample_layer.1 + ReLU
    (37, 29, 136, 264, 1, 1, 0, 3, True, False, 0),       # K tail (136 = 4 steps + 8), N = 256 + 8, SiLU
])
def test_gemm_h2(L, cfg, case):
    """fp32-equivalent: compared with an fp64 contraction of the UNROUNDED fp32 operands."""
    Hin, Win, Cin, N, k, stride, pad, act, use_bias, use_res, G = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(1, Cin, Hin, Win, generator=g) * 3.0
    w = torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(N, generator=g) * 0.1 if use_bias else None
    ref = F.conv2d(x.double(), w.double(), bias.double() if use_bias else None, stride=stride, padding=pad)
    mag = F.conv2d(x.abs().double(), w.abs().double(), None, stride=stride, padding=pad)       # sum |a||w|: error scale
    Hout, Wout = ref.shape[2:]
    M = Hout * Wout
    raw = ref.permute(0, 2, 3, 1).reshape(M, N)
    mag = mag.permute(0, 2, 3, 1).reshape(M, N)
    res = torch.randn(M, N, generator=g) if use_res else None
    exp = ACTS[act](raw) + (res.double() if use_res else 0)
    A = cast_h2(L, x.permute(0, 2, 3, 1).reshape(Hin * Win, Cin).contiguous().cuda())
    Wp, wscale = pack_weight_h2(L, w)
    outF = torch.full((M, N), float("nan"), device="cuda")
    want_b = N % 8 == 0
    outB = torch.zeros((M, N), device="cuda", dtype=torch.int32) if want_b else None
    stats = torch.zeros(64, device="cuda", dtype=torch.float64) if G else None
    bias_d = dev(bias) if use_bias else None
    res_d = dev(res) if use_res else None
    L.check(L.lib().uni_gemm_h2(L.ptr(A), Cin, L.ptr(Wp), wscale, M, N, Hin, Win, Cin, k, k, stride, pad,
                                L.ptr(bias_d), act, L.ptr(res_d), N, L.ptr(outF), N, L.ptr(outB), N,
                                L.ptr(stats), (N // G) if G else 0, cfg, L.stream_ptr()), "gemm_h2")
    torch.cuda.synchronize()
    got = outF.cpu().double()
    assert torch.isfinite(got).all()
    tol = mag * 2.0 ** -20 + 1e-6                     # operand split 2^-22 each side + fp32 accumulation; act slope <= ~1.1
    assert ((got - exp).abs() <= tol * 1.2).all(), ((got - exp).abs() / tol).max()
    if want_b:
        dec, _, _ = h2_decode(outB, M, N)
        assert ((dec.cpu().double() - exp).abs() <= tol * 1.2 + exp.abs() * 2.0 ** -21).all()
    if G:
        cpg = N // G
        grp = raw.reshape(M, G, cpg)
        s_ref = torch.stack([grp.sum((0, 2)), (grp ** 2).sum((0, 2))], 1)
        s_got = stats.cpu()[:2 * G].reshape(G, 2)
        assert torch.allclose(s_got, s_ref, rtol=2e-5, atol=2e-2), (s_got - s_ref).abs().max()    # sums of ~1e5 |terms| that nearly cancel


@pytest.mark.parametrize("case", [
    # (Hin, Win, Cin, N, k, stride, pad, act, bias, G, fp32_out): several 256 x 256 tiles per persistent block
    (200, 320, 64, 512, 3, 1, 1, 0, False, 16, True),      # head / FPN 3x3 + GroupNorm sums, 2 K steps per tap
    (101, 163, 96, 384, 3, 2, 1, 0, True, 16, True),       # 3x3 stride 2, odd map (ragged M, padding on every side), cpg 24
    (120, 160, 64, 256, 2, 2, 0, 0, True, 0, True),        # 2x2 / s2 downsample, no statistics
    (160, 200, 128, 256, 1, 1, 0, 0, False, 16, True),     # 1x1 lateral conv + GroupNorm sums (plain DMA path, block-wide reduction)
    (96, 96, 64, 256, 3, 1, 1, 1, True, 0, False),         # 3x3 + ReLU, operand-format output only (upsample_layer)
])
def test_gemm_h2q_conv_and_stats(L, case):
    """gemm_h2q.hip as an implicit GEMM (buffer-descriptor gather: the tap travels in the scalar offset, out-of-image lanes
    read zeros through the range check) and with the GroupNorm-statistics epilogue, against fp64 on the unrounded operands;
    three launches must agree bit for bit (the fp64 statistics atomics excepted)."""
    Hin, Win, Cin, N, k, stride, pad, act, use_bias, G, use_F = case
    g = torch.Generator().manual_seed(Hin * 7 + N)
    x = (torch.randn(1, Cin, Hin, Win, generator=g) * 2.0).cuda()
    w = (torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    bias = (torch.randn(N, generator=g) * 0.1).cuda() if use_bias else None
    ref = F.conv2d(x.double(), w.double().cuda(), bias.double() if use_bias else None, stride=stride, padding=pad)
    mag = F.conv2d(x.abs().double(), w.abs().double().cuda(), None, stride=stride, padding=pad)
    Hout, Wout = ref.shape[2:]
    M = Hout * Wout
    raw = ref.permute(0, 2, 3, 1).reshape(M, N)
    mag = mag.permute(0, 2, 3, 1).reshape(M, N)
    exp = ACTS[act](raw)
    A = cast_h2(L, x.permute(0, 2, 3, 1).reshape(Hin * Win, Cin).contiguous())
    Wp, wscale = pack_weight_h2(L, w)
    tol = mag * 2.0 ** -20 + 1e-6
    outs = []
    for rep in range(3):
        outF = torch.full((M, N), float("nan"), device="cuda") if use_F else None
        outB = torch.zeros((M, N), device="cuda", dtype=torch.int32) if not use_F else None
        stats = torch.zeros(64, device="cuda", dtype=torch.float64) if G else None
        L.check(L.lib().uni_gemm_h2(L.ptr(A), Cin, L.ptr(Wp), wscale, M, N, Hin, Win, Cin, k, k, stride, pad, L.ptr(bias), act, None, N,
                                    L.ptr(outF), N, L.ptr(outB), N, L.ptr(stats), (N // G) if G else 0, 188, L.stream_ptr()), "gemm_h2")
        torch.cuda.synchronize()
        got = outF.double() if use_F else h2_decode(outB, M, N)[0].double()
        assert torch.isfinite(got).all()
        assert ((got - exp).abs() <= tol * 1.2 + (0 if use_F else exp.abs() * 2.0 ** -21)).all(), ((got - exp).abs() / tol).max()
        outs.append(outF if use_F else outB)
        if G:
            grp = raw.reshape(M, G, N // G)
            s_ref = torch.stack([grp.sum((0, 2)), (grp ** 2).sum((0, 2))], 1)
            s_got = stats[:2 * G].reshape(G, 2)
            # fp32 partial sums per tile over ~1e6 O(1) terms that nearly cancel in the first column
            assert torch.allclose(s_got, s_ref, rtol=2e-5, atol=0.5), (s_got - s_ref).abs().max()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_gemm_h2_large_and_subnormal_lo(L):
    """256x256-tile path on a pwconv-sized problem + operands whose lo halves are f16 SUBNORMALS (|x| ~ 1e-2): the MFMA
    must not flush them (error would jump from ~2^-22 to ~2^-12 relative)."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 30001, 768, 192
    x = (torch.randn(M, K, generator=g) * 1e-2).cuda()
    w = torch.randn(N, K, generator=g) / K ** 0.5
    exp = x.double() @ w.double().cuda().t()
    mag = x.abs().double() @ w.abs().double().cuda().t()
    A = cast_h2(L, x)
    Wp, wscale = pack_weight_h2(L, w.reshape(N, K, 1, 1))
    for cfg in (0, 44, 22, 322, 323):
        outF = torch.full((M, N), float("nan"), device="cuda")
        L.check(L.lib().uni_gemm_h2(L.ptr(A), K, L.ptr(Wp), wscale, M, N, M, 1, K, 1, 1, 1, 0, None, 0, None, N,
                                    L.ptr(outF), N, None, N, None, 0, cfg, L.stream_ptr()), "gemm_h2")
        torch.cuda.synchronize()
        r = ((outF.double() - exp).abs() / (mag * 2.0 ** -20 + 1e-12)).max().item()
        assert r < 1.5, (cfg, r)


def test_wrapper_error_paths_raise_unicorn_error(L):
    """ops.postprocess_image / ops.letterbox reject wrong dtypes with UnicornHipError (callers catch that type)."""
    from unicorn_amd.ops import letterbox, postprocess_image
    with pytest.raises(L.UnicornHipError):
        postprocess_image(torch.zeros(10, 6, device="cuda", dtype=torch.float64), 1, 0.1, 0.5)
    with pytest.raises(L.UnicornHipError):
        letterbox(torch.zeros(8, 8, 3, device="cuda", dtype=torch.float32), (32, 32))


@pytest.mark.parametrize("cfg", [188, 44, 0])
@pytest.mark.parametrize("case", [
    # (M, N, K, act, res, outF, outB): several tiles per persistent block
    (70001, 256, 64, 2, False, False, True),      # two K steps per tile, GELU, operand-format out, ragged M
    (35003, 512, 160, 0, True, True, True),       # residual + fp32 + operand-format out
    (9001, 3072, 320, 2, False, False, True),     # pwconv1-like
    (20000, 768, 1024, 0, True, True, False),     # pwconv2-like, fp32 out only
    (5000, 136, 256, 1, False, True, True),       # N not a multiple of the tile (136 = 128 + 8), ReLU
])
def test_gemm_h2_persistent(L, cfg, case):
    """gemm_h2p.hip (cfg 144) and gemm_h2q.hip (cfg 188: ping-pong wave groups, counted-vmcnt DMA stream across K steps and
    tiles; K = 64 is its shortest stream, K = 160 an odd step count so the stage parity flips from tile to tile) against fp64 on
    the unrounded operands, next to the one-tile-per-block kernels.  The schedule-sensitive kernel is also run three times and
    must reproduce itself bit for bit (a DMA / fragment-read race shows up as run-to-run differences)."""
    M, N, K, act, use_res, use_F, use_B = case
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 2.0).cuda()
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    res = torch.randn(M, N, generator=g).cuda() if use_res else None
    raw = x.double() @ w.double().cuda().t() + bias.double()
    mag = x.abs().double() @ w.abs().double().cuda().t()
    exp = ACTS[act](raw) + (res.double() if use_res else 0)
    A = cast_h2(L, x)
    Wp, wscale = pack_weight_h2(L, w.reshape(N, K, 1, 1))
    outF = torch.full((M, N), float("nan"), device="cuda") if use_F else None
    outB = torch.zeros((M, N), device="cuda", dtype=torch.int32) if use_B else None
    L.check(L.lib().uni_gemm_h2(L.ptr(A), K, L.ptr(Wp), wscale, M, N, M, 1, K, 1, 1, 1, 0, L.ptr(bias), act, L.ptr(res), N,
                                L.ptr(outF), N, L.ptr(outB), N, None, 0, cfg, L.stream_ptr()), "gemm_h2")
    torch.cuda.synchronize()
    tol = mag * 2.0 ** -20 + 1e-6
    if use_F:
        assert torch.isfinite(outF).all()
        assert ((outF.double() - exp).abs() <= tol * 1.2).all(), ((outF.double() - exp).abs() / tol).max()
    if use_B:
        dec, _, _ = h2_decode(outB, M, N)
        assert ((dec.double() - exp).abs() <= tol * 1.2 + exp.abs() * 2.0 ** -21).all()
    if cfg == 188:
        first = (outF.clone() if use_F else None, outB.clone() if use_B else None)
        for _ in range(3):
            L.check(L.lib().uni_gemm_h2(L.ptr(A), K, L.ptr(Wp), wscale, M, N, M, 1, K, 1, 1, 1, 0, L.ptr(bias), act, L.ptr(res), N,
                                        L.ptr(outF), N, L.ptr(outB), N, None, 0, cfg, L.stream_ptr()), "gemm_h2")
            torch.cuda.synchronize()
            assert first[0] is None or torch.equal(first[0], outF)
            assert first[1] is None or torch.equal(first[1], outB)


# ------------------------------------------------------------------------------------------------
# mask post-processing on the device (row N1): integer / byte work -> BIT-EXACT against oracle/mask_oracle.py
# ------------------------------------------------------------------------------------------------
import mask_oracle as mo  # noqa: E402


@pytest.mark.parametrize("geo", [(80, 128, 0.5, 160, 256), (96, 160, 0.8333333, 110, 190), (100, 160, 1.37, 70, 100),
                                 (100, 160, 1.0, 100, 160), (50, 64, 0.7, 90, 40)])
def test_mask_resize_device(L, geo):
    from unicorn_amd.ops import mask_resize
    Hn, Wn, r, H, W = geo
    m = torch.rand(4, Hn, Wn, generator=torch.Generator().manual_seed(Hn + W))
    ref = mo.resize_bilinear(m.numpy(), r, H, W)
    got = mask_resize(m.cuda(), r, H, W).cpu().numpy()
    assert np.array_equal(got, ref)                                            # same fp32 operation order, no FMA contraction
    assert np.array_equal(mask_resize(m.cuda(), r, H, W, thr=0.3).cpu().numpy(), (ref > np.float32(0.3)).astype(np.uint8))
    assert mask_resize(m[:0].cuda(), r, H, W).shape == (0, H, W)


def test_vos_merge_device(L):
    from unicorn_amd.ops import vos_merge
    g = np.random.default_rng(3)
    Hn, Wn, r, H, W = 100, 160, 0.75, 130, 200
    probs = g.random((4, Hn, Wn), dtype=np.float32)
    probs[0, :20] = 0.0
    probs[1, 20:40] = 1.0                                                      # background exactly 0 there
    probs[2] = probs[3]                                                        # exact ties between two objects: the lower id wins
    ids = ["4", "2", "9", "6"]
    init = (g.random((2, H, W)) > 0.8).astype(np.uint8)
    ref = mo.soft_aggregate(mo.resize_bilinear(probs, r, H, W), ids, init, ["11", "1"])
    got = vos_merge(torch.from_numpy(probs).cuda(), ids, r, H, W, torch.from_numpy(init).cuda(), ["11", "1"]).cpu().numpy()
    assert np.array_equal(got, ref)
    ref2 = mo.soft_aggregate(mo.resize_bilinear(probs, r, H, W), ids)
    assert np.array_equal(vos_merge(torch.from_numpy(probs).cuda(), ids, r, H, W).cpu().numpy(), ref2)


def test_vos_mots_device_vs_reference_line_goldens(L, golden_dir):
    """the device kernels against outputs of the reference's OWN lines (tests/golden/make_golden_vos.py execs
    unicorn_vos.py:99-120 and mot_evaluator.py:804-805 / 860-865): bit-identical id maps, thresholded masks, overlap-free masks"""
    from unicorn_amd.ops import vos_merge, mots_overlap_free
    from unicorn_amd.utils.masks import mots_threshold
    g = np.load(os.path.join(golden_dir, "vos_mots_ref.npz"))
    for tag in "abcd":
        probs, ids = g["vos_%s_probs" % tag], [str(k) for k in g["vos_%s_ids" % tag]]
        init, init_ids = g["vos_%s_init" % tag], [str(k) for k in g["vos_%s_init_ids" % tag]]
        ref = g["vos_%s_final" % tag]
        H, W = ref.shape
        got = vos_merge(torch.from_numpy(probs).cuda(), ids, 1.0, H, W,
                        torch.from_numpy(init).cuda() if len(init_ids) else None, init_ids).cpu().numpy()
        assert np.array_equal(got, ref), tag
        prob = g["mots_%s_prob" % tag]
        scale, img_h, img_w = g["mots_%s_geom" % tag]
        shape = tuple(int(v) for v in g["mots_%s_shape" % tag])
        n = int(np.prod(shape))
        mref = np.unpackbits(g["mots_%s_masks" % tag])[:n].reshape(shape)
        fref = np.unpackbits(g["mots_%s_free" % tag])[:n].reshape(shape)
        m = mots_threshold(torch.from_numpy(prob)[:, None].cuda(), float(scale), int(img_h), int(img_w), 0.5)
        assert tuple(m.shape) == shape and np.array_equal(m.cpu().numpy(), mref), tag
        assert np.array_equal(mots_overlap_free(m).cpu().numpy(), fref), tag


def test_mots_overlap_free_and_rle_device(L):
    from unicorn_amd.ops import mots_overlap_free, rle_encode
    g = np.random.default_rng(4)
    H, W = 135, 241
    masks = np.zeros((6, H, W), dtype=np.uint8)
    for n in range(5):
        y0, x0 = g.integers(0, H - 40), g.integers(0, W - 60)
        masks[n, y0:y0 + g.integers(10, 40), x0:x0 + g.integers(10, 60)] = 1
    masks[3] |= (g.random((H, W)) < 0.05).astype(np.uint8)                     # speckle: many short runs
    masks[4, 0, 0] = 1                                                         # first column-major element set: leading zero-length run
    # masks[5] stays empty: one run of h*w zeros
    of = mots_overlap_free(torch.from_numpy(masks).cuda())
    assert np.array_equal(of.cpu().numpy(), mo.overlap_free(masks))
    strs = rle_encode(of)
    exp = [mo.mask_to_rle_string(m) for m in mo.overlap_free(masks)]
    assert strs == exp
    for s, m in zip(strs, mo.overlap_free(masks)):                             # and they decode back through the rleFrString restatement
        assert np.array_equal(mo.rle_decode(mo.rle_from_string(s), H, W), m)
    assert rle_encode(of[:0]) == []
    dense = (g.random((2, 64, 96)) < 0.5).astype(np.uint8)                      # > max_runs runs: the wrapper retries with larger bounds
    assert rle_encode(torch.from_numpy(dense).cuda(), max_runs=64) == [mo.mask_to_rle_string(m) for m in dense]


def test_mots_mask_pipeline_device(L):
    """mot_evaluator.py:804-805 + :850-863 + :889-892 composed (unicorn_amd/utils/masks.py) vs the oracle composition"""
    from unicorn_amd.utils.masks import mots_rle, mots_threshold
    g = torch.Generator().manual_seed(9)
    Hn, Wn, scale, h, w = 100, 160, 0.74, 130, 210
    score = torch.rand(5, 1, Hn, Wn, generator=g)
    score[2, 0, 20:70, 30:120] = 0.9
    score[0, 0, 40:90, 60:150] = 0.8
    m_dev = mots_threshold(score.cuda(), scale, h, w, 0.30)
    m_ref = (mo.resize_bilinear(score[:, 0].numpy(), scale, h, w) > np.float32(0.30)).astype(np.uint8)
    assert np.array_equal(m_dev.cpu().numpy(), m_ref)
    order = [3, 0, 2]                                                  # association kept 3 of 5, ascending track id order
    free, strs = mots_rle(m_dev, order)
    ref_free = mo.overlap_free(m_ref[order])
    assert np.array_equal(free.cpu().numpy(), ref_free)
    assert strs == [mo.mask_to_rle_string(m).decode("utf-8") for m in ref_free]
    assert mots_rle(m_dev[:0])[1] == [] and mots_threshold(score[:0].cuda(), scale, h, w).shape == (0, h, w)


# ------------------------------------------------------------------------------------------------
# fused ConvNeXt MLP (mlp_fused.hip): pwconv1 -> GELU -> pwconv2 (gamma folded) -> + residual in one launch
# ------------------------------------------------------------------------------------------------
def mlp_pack(L, w1, w2, gamma, layout=0):
    C_ = w1.shape[1]
    nb = L.lib().uni_mlp_blob_bytes(C_)
    assert nb == 32 * C_ * C_
    blob = np.zeros(nb // 2, dtype=np.uint16)
    a, b = C.c_float(0), C.c_float(0)
    w1c, w2c, gc = (np.ascontiguousarray(t.float().numpy()) for t in (w1, w2, gamma))
    L.check(L.lib().uni_mlp_pack(w1c.ctypes.data_as(C.c_void_p), w2c.ctypes.data_as(C.c_void_p), gc.ctypes.data_as(C.c_void_p), C_, layout,
                                 blob.ctypes.data_as(C.c_void_p), C.byref(a), C.byref(b)), "mlp_pack")
    return torch.from_numpy(blob.view(np.int16)).cuda(), a.value, b.value


@pytest.mark.parametrize("with_outb", [False, True])
@pytest.mark.parametrize("C_,M,layout", [(96, 1000, 0), (192, 128, 0), (192, 33000, 0), (256, 4000, 0), (256, 40001, 0), (192, 1, 0), (96, 70000, 0),
                                         (192, 128, 1), (192, 33000, 1), (256, 4000, 1), (256, 40001, 1), (192, 1, 1), (192, 77777, 1)])
def test_mlp_fused(L, C_, M, layout, with_outb):
    """convnext.py:47-54 after the LayerNorm: x + gamma * (W2 GELU(W1 a + b1) + b2), against torch fp64 on the f16x2-decoded
    operand; ragged M (rows past M are neither read nor written), several tiles per block (M > 128 * 256), in-place residual."""
    g = torch.Generator().manual_seed(C_ + M)
    x = torch.randn(M, C_, generator=g) * 1.5
    w1 = torch.randn(4 * C_, C_, generator=g) * 0.05
    b1 = torch.randn(4 * C_, generator=g) * 0.2
    w2 = torch.randn(C_, 4 * C_, generator=g) * 0.05
    b2 = torch.randn(C_, generator=g) * 0.2
    gamma = torch.rand(C_, generator=g) + 0.5
    res = torch.randn(M, C_, generator=g) * 3.0
    A = cast_h2(L, x.cuda())
    a_dec = h2_decode(A, M, C_)[0].cpu().double()
    hid = F.gelu(a_dec @ w1.double().t() + b1.double())
    ref = res.double() + gamma.double() * (hid @ w2.double().t() + b2.double())
    blob, ws1, ws2 = mlp_pack(L, w1, w2, gamma, layout)
    pad = 64                                                 # guard rows behind the output: must stay untouched
    out = torch.full((M + pad, C_), 777.0, device="cuda")
    out[:M] = res.cuda()
    outb = torch.zeros((M + pad, C_), device="cuda", dtype=torch.int32) if with_outb else None
    b1d, b2d = b1.cuda(), (gamma * b2).cuda()
    L.check(L.lib().uni_mlp_fused(L.ptr(A), C_, L.ptr(blob), L.ptr(b1d), L.ptr(b2d), ws1, ws2, L.ptr(out), C_,
                                  L.ptr(out), C_, L.ptr(outb), C_, M, C_, layout, 0, L.stream_ptr()), "mlp_fused")
    torch.cuda.synchronize()
    got = out[:M].cpu().double()
    scale = max(1.0, ref.abs().max().item())
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() < 4e-6 * scale, (got - ref).abs().max()
    assert (out[M:] == 777.0).all()
    if with_outb:
        dec = h2_decode(outb[:M], M, C_)[0].cpu().double()
        assert (dec - got).abs().max() < 1e-6 * scale
        assert (outb[M:] == 0).all()


def test_mlp_fused_matches_unfused_pair(L):
    """Same block through the two launches the engine used before (uni_gemm_h2: pwconv1 + GELU -> f16x2 hidden, pwconv2 + residual):
    the fused kernel rounds the hidden activations to the same f16x2 format, so both sit within fp32 round-off of each other."""
    C_, M = 192, 5000
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, C_, generator=g)
    w1 = torch.randn(4 * C_, C_, generator=g) * 0.04
    b1 = torch.randn(4 * C_, generator=g) * 0.1
    w2 = torch.randn(C_, 4 * C_, generator=g) * 0.04
    b2 = torch.randn(C_, generator=g) * 0.1
    gamma = torch.rand(C_, generator=g) + 0.5
    res = torch.randn(M, C_, generator=g).cuda()
    A = cast_h2(L, x.cuda())
    blob, ws1, ws2 = mlp_pack(L, w1, w2, gamma)
    fused = res.clone()
    b1d, b2d = b1.cuda(), (gamma * b2).cuda()
    L.check(L.lib().uni_mlp_fused(L.ptr(A), C_, L.ptr(blob), L.ptr(b1d), L.ptr(b2d), ws1, ws2, L.ptr(fused), C_,
                                  L.ptr(fused), C_, None, 0, M, C_, 0, 0, L.stream_ptr()), "mlp_fused")
    W1p, s1 = pack_weight_h2(L, w1.reshape(4 * C_, C_, 1, 1))
    W2p, s2 = pack_weight_h2(L, (gamma[:, None] * w2).reshape(C_, 4 * C_, 1, 1))
    hid = torch.zeros((M, 4 * C_), device="cuda", dtype=torch.int32)
    L.check(L.lib().uni_gemm_h2(L.ptr(A), C_, L.ptr(W1p), s1, M, 4 * C_, M, 1, C_, 1, 1, 1, 0, L.ptr(b1d), 2, None, 0, None, 0,
                                L.ptr(hid), 4 * C_, None, 0, 0, L.stream_ptr()), "pw1")
    two = torch.zeros((M, C_), device="cuda")
    L.check(L.lib().uni_gemm_h2(L.ptr(hid), 4 * C_, L.ptr(W2p), s2, M, C_, M, 1, 4 * C_, 1, 1, 1, 0, L.ptr(b2d), 0, L.ptr(res), C_,
                                L.ptr(two), C_, None, 0, None, 0, 0, L.stream_ptr()), "pw2")
    torch.cuda.synchronize()
    assert (fused - two).abs().max().item() < 3e-6 * max(1.0, two.abs().max().item())


def test_mots_threshold_matches_reference_crop_shape():
    """mot_evaluator.py:804-805 on a 480 x 854 image at 800 x 1280: F.interpolate(scale_factor=1/scale) yields 480 x 853, so the
    reference thresholds and RLE-encodes a (480, 853) mask.  mots_threshold(crop=True) returns that shape with identical bits;
    crop=False is the zero-padded full-size map of the VOS driver."""
    from unicorn_amd.utils.masks import mots_threshold
    img_h, img_w, Hn, Wn = 480, 854, 800, 1280
    scale = min(Hn / float(img_h), Wn / float(img_w))
    g = torch.Generator().manual_seed(0)
    om = torch.rand(3, 1, Hn, Wn, generator=g)
    ref = F.interpolate(om, scale_factor=1 / scale, mode="bilinear", align_corners=False)[:, 0, :img_h, :img_w] > 0.3
    assert ref.shape[2] == img_w - 1                                  # the case the reference silently shortens
    got = mots_threshold(om.cuda(), scale, img_h, img_w, 0.3)
    assert tuple(got.shape) == tuple(ref.shape)
    assert (got.cpu().bool() != ref).float().mean() < 1e-5            # thresholding an fp32 interpolation: ties at round-off only
    full = mots_threshold(om.cuda(), scale, img_h, img_w, 0.3, crop=False)
    assert tuple(full.shape) == (3, img_h, img_w) and not full[:, :, img_w - 1:].any()


@pytest.mark.parametrize("cfg", [322, 323, 332, 331, 346, 422, 423])      # 422 / 423: 322 / 323 with producer waves (round 5)
@pytest.mark.parametrize("case", [
    # (Hin, Win, Cin, N, k, stride, pad, act, res, stats_G, outB, splitk)
    (4000, 1, 3072, 768, 1, 1, 0, 0, True, 0, False, 1),      # stage-2 pwconv2 of one frame + in-place residual (256 tiles of 128 x 96)
    (1003, 1, 768, 3072, 1, 1, 0, 2, False, 0, True, 1),      # pwconv1 + GELU -> operand-format out, ragged M
    (50, 80, 384, 384, 3, 1, 1, 0, False, 16, False, 1),      # FPN / head 3x3, zero padding through the range check, GroupNorm sums
    (26, 34, 192, 192, 3, 2, 1, 0, False, 16, False, 1),      # 3x3 stride 2, ragged M
    (20, 20, 96, 200, 2, 2, 0, 1, False, 0, True, 1),         # 2x2 / s2 downsample, ReLU, N = 200 (ragged in every tile width)
    (130, 1, 64, 96, 1, 1, 0, 0, False, 0, False, 1),         # K = 64: two slices, fewer than the ring holds
    (25, 40, 256, 256, 3, 1, 1, 0, False, 16, False, 4),      # split-K: K ranges of 18 slices, statistics from the reduce kernel
    (1000, 1, 3072, 768, 1, 1, 0, 0, True, 0, False, 5),      # split-K with unequal ranges (96 slices / 5) + in-place residual
])
def test_gemm_h2_deep(L, cfg, case):
    """gemm_h2d.hip (4-wave tiles with an NST-deep counted-vmcnt DMA ring, descriptor addressing) against fp64 on the unrounded
    operands AND bit for bit against gemm_h2_kernel (cfg 22: same MFMA order per accumulator, so fp32 results must be identical);
    three launches must reproduce themselves (a DMA / fragment-read race shows up as run-to-run differences)."""
    Hin, Win, Cin, N, k, stride, pad, act, use_res, G, use_B, sk = case
    g = torch.Generator().manual_seed(Hin * 3 + N + k)
    x = (torch.randn(1, Cin, Hin, Win, generator=g) * 2.0).cuda()
    w = torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    ref = F.conv2d(x.double(), w.double().cuda(), bias.double(), stride=stride, padding=pad)
    mag = F.conv2d(x.abs().double(), w.abs().double().cuda(), None, stride=stride, padding=pad)
    Hout, Wout = ref.shape[2:]
    M = Hout * Wout
    raw = ref.permute(0, 2, 3, 1).reshape(M, N)
    mag = mag.permute(0, 2, 3, 1).reshape(M, N)
    res0 = torch.randn(M, N, generator=g).cuda() if use_res else None
    exp = ACTS[act](raw) + (res0.double() if use_res else 0)
    A = cast_h2(L, x.permute(0, 2, 3, 1).reshape(Hin * Win, Cin).contiguous())
    Wp, wscale = pack_weight_h2(L, w)
    tol = mag * 2.0 ** -20 + 1e-6

    def run(c):
        outF = res0.clone() if use_res else (None if use_B else torch.full((M, N), float("nan"), device="cuda"))
        outB = torch.zeros((M, N), device="cuda", dtype=torch.int32) if use_B else None
        stats = torch.zeros(64, device="cuda", dtype=torch.float64) if G else None
        L.check(L.lib().uni_gemm_h2(L.ptr(A), Cin, L.ptr(Wp), wscale, M, N, Hin, Win, Cin, k, k, stride, pad, L.ptr(bias), act,
                                    L.ptr(outF) if use_res else None, N, L.ptr(outF), N, L.ptr(outB), N, L.ptr(stats), (N // G) if G else 0,
                                    c + (1000000 * sk if sk > 1 else 0), L.stream_ptr()), "gemm_h2")
        torch.cuda.synchronize()
        return (outB if use_B else outF), stats

    outs = [run(cfg) for _ in range(3)]
    got = h2_decode(outs[0][0], M, N)[0].double() if use_B else outs[0][0].double()
    assert torch.isfinite(got).all()
    assert ((got - exp).abs() <= tol * 1.2 + (exp.abs() * 2.0 ** -21 if use_B else 0)).all(), ((got - exp).abs() / tol).max()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0])
    base, bstats = run(22)
    assert torch.equal(outs[0][0], base)
    if G:
        grp = raw.reshape(M, G, N // G)
        s_ref = torch.stack([grp.sum((0, 2)), (grp ** 2).sum((0, 2))], 1)
        s_got = outs[0][1][:2 * G].reshape(G, 2)
        assert torch.allclose(s_got, s_ref, rtol=2e-5, atol=5e-2), (s_got - s_ref).abs().max()


@pytest.mark.parametrize("case", [
    # (Hin, Win, Cin, N, k, res, stats_G, splitk, cfg)
    (25, 40, 256, 256, 3, False, 16, 4, 0),          # head tower conv at stride 32, GroupNorm sums from the reduce kernel
    (50, 80, 384, 384, 3, False, 16, 2, 22),
    (1000, 1, 3072, 768, 1, True, 0, 3, 0),          # pwconv2 + in-place residual, ragged M, K ranges of unequal length
    (130, 1, 1536, 64, 1, False, 0, 8, 0),           # N = 64: 128 x 64 tiles
    (4000, 1, 3072, 768, 1, True, 0, 4, 188),        # stage-2 pwconv2 of one frame: K ranges as work units of the ping-pong kernel
    (100, 160, 256, 256, 3, False, 16, 4, 188),      # head tower 3x3 conv, ping-pong implicit GEMM, statistics in the reduce
    (25, 40, 768, 768, 3, False, 16, 6, 0),          # heuristic takes the ping-pong kernel (K steps divisible by the split)
    (50, 80, 384, 384, 3, False, 16, 3, 188),        # ragged M tile (4000 rows), 108 K steps / 3
])
def test_gemm_h2_splitk(L, case):
    """Split-K of the single-frame path (gemm_h2.hip): K ranges -> partial-tile slab -> splitk_reduce_kernel (bias, residual, GroupNorm
    sums) must equal the unsplit launch of the same problem to fp32 round-off, and torch fp64 to the f16x2 bound."""
    Hin, Win, Cin, N, k, use_res, G, sk, cfg = case
    g = torch.Generator().manual_seed(Hin + N + sk)
    pad = (k - 1) // 2
    M, K = Hin * Win, Cin * k * k
    x = torch.randn(1, Cin, Hin, Win, generator=g)
    w = torch.randn(N, Cin, k, k, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    A = cast_h2(L, x[0].permute(1, 2, 0).reshape(M, Cin).contiguous().cuda())
    Wp, ws = pack_weight_h2(L, w)
    res0 = torch.randn(M, N, generator=g).cuda()
    bias_d = bias.cuda()
    outs, stats = [], []
    for split in (1, sk):
        out = res0.clone() if use_res else torch.zeros((M, N), device="cuda")
        st = torch.zeros(64, device="cuda", dtype=torch.float64) if G else None
        L.check(L.lib().uni_gemm_h2(L.ptr(A), Cin, L.ptr(Wp), ws, M, N, Hin, Win, Cin, k, k, 1, pad, L.ptr(bias_d), 0,
                                    L.ptr(out) if use_res else None, N, L.ptr(out), N, None, 0, L.ptr(st), N // G if G else 0,
                                    cfg + (1000000 * split if split > 1 else 0), L.stream_ptr()), "gemm_h2")
        torch.cuda.synchronize()
        outs.append(out.cpu().double())
        stats.append(None if st is None else st.cpu())
    ref = F.conv2d(h2_decode(A, M, Cin)[0].cpu().double().reshape(Hin, Win, Cin).permute(2, 0, 1)[None], w.double(), bias.double(), padding=pad)
    ref = ref[0].permute(1, 2, 0).reshape(M, N) + (res0.cpu().double() if use_res else 0)
    scale = max(1.0, ref.abs().max().item())
    tol = 4e-6 if K <= 4096 else 1e-5                     # fp32 accumulation over K terms
    assert (outs[1] - ref).abs().max() < tol * scale
    assert (outs[1] - outs[0]).abs().max() < tol * scale
    if G:
        assert (stats[1][:2 * G] - stats[0][:2 * G]).abs().max() < 1e-5 * stats[0][:2 * G].abs().max()


# ------------------------------------------------------------------------------------------------
# independent sanity bounds for the UNPINNED third-party restatements: the device kernels against
# tests/sanity_refs.py -- second statements written by a different route than oracle/*.py (CPU half: tests/test_sanity_bounds_cpu.py)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,size,swap", [((1080, 1920), (800, 1280), True), ((480, 640), (800, 1280), True),
                                             ((375, 1242), (800, 1280), False)])
def test_letterbox_device_within_one_lsb_of_float_bilinear(L, shape, size, swap):
    """uni_letterbox (OpenCV's 11-bit fixed point) vs torch's FLOAT bilinear on the same uint8 image: never more than one LSB apart,
    >= 90 % identical after rounding, pad value exact; a smooth ramp exposes coordinate / orientation errors."""
    import sanity_refs as sr
    from unicorn_amd.ops import letterbox
    g = np.random.default_rng(shape[0] + shape[1])
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    ramp = np.stack([(xx * 255.0 / shape[1]), (yy * 255.0 / shape[0]), ((xx + yy) * 255.0 / (shape[0] + shape[1]))], -1).astype(np.uint8)
    for img in (g.integers(0, 256, shape + (3,), dtype=np.uint8), ramp):
        out, r = letterbox(img, size, swap_rb=swap)
        ref, r2, (rh, rw) = sr.float_letterbox(img, size, swap)
        o = out[0].cpu().numpy().astype(np.float64)
        assert r == r2 and np.abs(o - ref).max() <= 1.0
        assert (o == np.rint(ref)).mean() > 0.9
        assert (o[:, rh:] == 114).all() and (o[:, :, rw:] == 114).all()


def test_nms_device_vs_brute_force_adversarial(L):
    """uni_nms against scalar O(N^2) greedy loops written from the torchvision docstring: IoU exactly at the threshold (strict >),
    equal scores (lower index first), class offsets of the coordinate trick near the fp32 resolution, a dense field"""
    import sanity_refs as sr
    from unicorn_amd.utils.boxes import batched_nms, nms
    for name, b, s, c, thr in sr.adversarial_nms_cases():
        tb, ts, tc = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(np.asarray(c, np.int64)).cuda()
        assert nms(tb, ts, thr).cpu().tolist() == sr.brute_nms(b, s, thr), name
        assert batched_nms(tb, ts, tc, thr).cpu().tolist() == sr.brute_batched_nms(b, s, c, thr), name


def test_rle_device_strings_decode_through_the_independent_reader(L):
    """uni_rle_encode strings of 1-pixel, all-ones, column- / row-alternating and empty masks through a character-level reader of the
    documented pycocotools format (not the oracle's codec)"""
    import sanity_refs as sr
    from unicorn_amd.ops import rle_encode
    cases = sr.rle_sanity_masks()
    for shape in sorted({m.shape for _, m in cases}):
        group = [(n, m) for n, m in cases if m.shape == shape]
        strs = rle_encode(torch.from_numpy(np.stack([m for _, m in group])).cuda())
        for (n, m), s in zip(group, strs):
            assert np.array_equal(sr.coco_rle_string_to_mask(s, *shape), m), (n, shape, s)
