"""VQ-VAE image tokenizer: module surface of vqvae/vqvae_zc.py (class names, constructor arguments,
state_dict keys: enc_b.blocks.{0,2,4,6}.*, quantize_t.{embed,cluster_size,embed_avg}, dec.blocks.{0,2,4,6}.*),
inference arithmetic in the HIP library (cogview_amd/csrc/conv.hip).

Scope: the frozen tokenizer as CogView uses it (vqvae/api.py: img2code / code2img, eval mode).  The production topology
-- stride=6, simple=True, n_res_block=0: three 4x4 stride-2 convolutions + 1x1, nearest-code search, three 4x4 stride-2
transposed convolutions + 1x1 -- runs the fused fast path (final 1x1 -> RGB in the last transposed convolution's epilogue).
Every OTHER topology the reference's constructors can build (vqvae/vqvae_zc.py:116-214: stride 6 / 4 / 2, simple or not,
any number of ResBlocks) runs through a generic interpreter of the same block list on the same kernels (round 4: 3x3
convolution, input ReLU and residual add in the convolution kernel) -- channel counts a multiple of 8.  Training of the VQ-VAE (EMA codebook update, Gumbel-softmax relaxation) is not
part of CogView's pipeline (no training script in the reference) and raises.
"""
import ctypes as C

import torch
from torch import nn

from .. import _lib as L
from .. import ops


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _conv(kind, x, w, bias, cout, relu, rgb=None, relu_in=False, residual=None, relu_residual=False):
    """x NHWC fp32 contiguous -> NHWC fp32.  rgb = (w_rgb [3, cout], bias_rgb [3], scale, shift): the decoder's final
    1x1 convolution (and the de-normalisation of api.code2img) fused into this layer's epilogue -- the layer's own
    output tensor is never written; returns the NCHW image [b, 3, oh, ow]."""
    if not x.is_cuda:
        raise L.CogviewHipError("cogview_amd.vqvae runs only on an MI355X (HIP) device; there is no CPU fallback")
    b, ih, iw, cin = x.shape
    if kind == L.CONV_4X4_S2:
        oh, ow = ih // 2, iw // 2
    elif kind == L.CONVT_4X4_S2:
        oh, ow = ih * 2, iw * 2
    else:
        oh, ow = ih, iw
    d = L.ConvDesc()
    d.kind, d.B, d.IH, d.IW, d.Cin, d.Cout, d.relu = kind, b, ih, iw, cin, cout, int(relu)
    d.relu_in, d.relu_residual = int(relu_in), int(relu_residual)
    if residual is not None:
        assert residual.shape == (b, oh, ow, cout) and residual.is_contiguous() and rgb is None
        d.residual = residual.data_ptr()
    setattr(d, "in", x.data_ptr())
    d.w, d.bias = w.data_ptr(), bias.data_ptr()
    if rgb is None:
        out = torch.empty((b, oh, ow, cout), dtype=torch.float32, device=x.device)
        d.out = out.data_ptr()
    else:
        ntiles = cout // 128
        out = torch.empty((ntiles, b * oh * ow, 4), dtype=torch.float32, device=x.device)
        d.rgb_w, d.rgb_partial = rgb[0].data_ptr(), out.data_ptr()
    par = 4 if kind == L.CONVT_4X4_S2 else 1
    taps = {L.CONV_4X4_S2: 16, L.CONV_1X1: 1, L.CONVT_4X4_S2: 4, L.CONV_3X3_S1: 9}[kind]
    npix_out = b * oh * ow
    with ops.timed_launch("conv", 2.0 * npix_out * cout * taps * cin, 4.0 * (x.numel() + w.numel() + out.numel()),
                          f"kind{kind} {b}x{ih}x{iw}x{cin}->{cout}"):
        L.check(L.lib().cogv_conv2d_nhwc_f32(C.byref(d), _stream()), "cogv_conv2d_nhwc_f32")
    if rgb is not None:
        img = torch.empty((b, 3, oh, ow), dtype=torch.float32, device=x.device)
        sc = (C.c_float * 3)(*rgb[2]) if rgb[2] is not None else None
        sh = (C.c_float * 3)(*rgb[3]) if rgb[3] is not None else None
        with ops.timed_launch("rgb_finalize", 0.0, 4.0 * (out.numel() + img.numel()), f"{b}x{oh}x{ow}"):
            L.check(L.lib().cogv_rgb_finalize_f32(_p(out), cout // 128, _p(rgb[1]), _p(img), b, oh, ow, sc, sh, _stream()),
                    "cogv_rgb_finalize_f32")
        return img
    return out


def pack_conv_weight(w):
    """Conv2d weight [Cout, Cin, kh, kw] -> [Cout, kh*kw, Cin4] (Cin padded to a multiple of 4)."""
    co, ci, kh, kw = w.shape
    ci4 = (ci + 3) // 4 * 4
    out = torch.zeros((co, kh * kw, ci4), dtype=torch.float32, device=w.device)
    out[:, :, :ci] = w.detach().float().permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
    return out.contiguous()


def pack_convt_weight(w):
    """ConvTranspose2d weight [Cin, Cout, 4, 4] (stride 2, pad 1) -> [4 parities][Cout][4 taps][Cin]:
    output pixel (2y+py, 2x+px) = sum over taps (ty,tx) of in[y+offy, x+offx] with ky = 2*ty (py=1) or 1+2*ty (py=0)."""
    ci, co, kh, kw = w.shape
    assert kh == 4 and kw == 4
    wf = w.detach().float()
    out = torch.empty((4, co, 4, ci), dtype=torch.float32, device=w.device)
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    ky = 2 * ty if py else 1 + 2 * ty
                    kx = 2 * tx if px else 1 + 2 * tx
                    out[py * 2 + px, :, ty * 2 + tx, :] = wf[:, :, ky, kx].t()
    return out.contiguous()


class Quantize(nn.Module):
    """Codebook holder (vqvae/vqvae_zc.py:26-96).  `embed` is [dim, n_embed] as in the reference."""

    def __init__(self, dim, n_embed, decay=0.99, eps=1e-5):
        super().__init__()
        self.dim, self.n_embed, self.decay, self.eps = dim, n_embed, decay, eps
        embed = torch.randn(dim, n_embed)
        torch.nn.init.xavier_uniform_(embed, gain=torch.nn.init.calculate_gain('tanh'))
        self.register_buffer("embed", embed)
        self.register_buffer("cluster_size", torch.zeros(n_embed))
        self.register_buffer("embed_avg", embed.clone())
        self._cache = None

    def _tables(self):
        """E^T [n_embed, dim] and |E_j|^2, rebuilt when the buffer changes (one-time set-up, not the hot path)."""
        key = (self.embed.data_ptr(), self.embed._version)
        if self._cache is None or self._cache[0] != key:
            et = self.embed.detach().float().t().contiguous()
            e2 = self.embed.detach().float().pow(2).sum(0).contiguous()       # vqvae_zc.py:46 `embed.pow(2).sum(0)`
            self._cache = (key, et, e2)
        return self._cache[1], self._cache[2]

    def nearest_code(self, x_nhwc):
        """x [b, h, w, dim] fp32 -> ids [b, h, w] int64 (argmax of -dist, first maximum)."""
        et, e2 = self._tables()
        flat = x_nhwc.reshape(-1, self.dim)
        ids = torch.empty(flat.shape[0], dtype=torch.int64, device=flat.device)
        with ops.timed_launch("vq_argmin", 2.0 * flat.shape[0] * self.dim * self.n_embed,
                              4.0 * (flat.numel() + et.numel()) + 8.0 * flat.shape[0], f"{flat.shape[0]}x{self.dim}x{self.n_embed}"):
            L.check(L.lib().cogv_vq_argmin_f32(_p(flat), _p(et), _p(e2), _p(ids), flat.shape[0], self.dim, self.n_embed,
                                               _stream()), "cogv_vq_argmin_f32")
        return ids.view(*x_nhwc.shape[:-1])

    def embed_code(self, embed_id):
        """F.embedding(ids, embed^T) -> [..., dim] (vqvae/vqvae_zc.py:95-96)."""
        et, _ = self._tables()
        ids = embed_id.contiguous()
        out = torch.empty(tuple(ids.shape) + (self.dim,), dtype=torch.float32, device=et.device)
        L.check(L.lib().cogv_embed_code_f32(_p(ids), _p(et), _p(out), ids.numel(), self.dim, self.n_embed, _stream()),
                "cogv_embed_code_f32")
        return out

    def forward_(self, input, continuous_relax=False, temperature=1., hard=False):
        if continuous_relax or self.training:
            raise NotImplementedError("only the frozen eval-mode tokenizer path is implemented (argmax + lookup)")
        ids = self.nearest_code(input.contiguous())
        quantize = self.embed_code(ids)
        diff = (quantize - input).pow(2).mean()
        return quantize, diff, ids


class _ConvStack(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.Sequential(*blocks)
        self._packed = None

    def _weights(self, packers):
        key = tuple((m.weight.data_ptr(), m.weight._version) for m in self.blocks if hasattr(m, "weight"))
        if self._packed is None or self._packed[0] != key:
            mods = [m for m in self.blocks if hasattr(m, "weight")]
            self._packed = (key, [(pk(m.weight), m.bias.detach().float().contiguous()) for pk, m in zip(packers, mods)])
        return self._packed[1]


class ResBlock(nn.Module):
    """vqvae/vqvae_zc.py:99-114: ReLU(inplace), conv3x3, ReLU(inplace), conv1x1, `out += input`.  The leading in-place ReLU
    overwrites the block's input before it is added back, so the block computes conv(...) + relu(input)."""

    def __init__(self, in_channel, channel):
        super().__init__()
        self.conv = nn.Sequential(nn.ReLU(inplace=True), nn.Conv2d(in_channel, channel, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(channel, in_channel, 1))


def _pad8(n):
    return (n + 7) // 8 * 8


def _pack_generic(m):
    """(kind, packed weight [.., Cout8, taps, Cin4], bias [Cout8], Cout) of one convolution module; output channels are padded
    to a multiple of 8 with zero filters (the kernel's 8-wide output vectors), input channels to a multiple of 4."""
    w = m.weight.detach().float()
    if isinstance(m, nn.ConvTranspose2d):
        if m.kernel_size == (1, 1):                       # a 1x1 transposed convolution is a 1x1 convolution with W^T
            kind, wp = L.CONV_1X1, pack_conv_weight(w.permute(1, 0, 2, 3))
        else:
            assert m.kernel_size == (4, 4) and m.stride == (2, 2) and m.padding == (1, 1)
            kind, wp = L.CONVT_4X4_S2, pack_convt_weight(w)
            if wp.shape[3] % 4:
                wp = torch.nn.functional.pad(wp, (0, 4 - wp.shape[3] % 4))
    else:
        ks, st, pd = m.kernel_size, m.stride, m.padding
        if ks == (4, 4) and st == (2, 2) and pd == (1, 1):
            kind = L.CONV_4X4_S2
        elif ks == (3, 3) and st == (1, 1) and pd == (1, 1):
            kind = L.CONV_3X3_S1
        elif ks == (1, 1) and st == (1, 1) and pd == (0, 0):
            kind = L.CONV_1X1
        else:
            raise NotImplementedError(f"convolution {ks} stride {st} padding {pd} does not occur in vqvae/vqvae_zc.py")
        wp = pack_conv_weight(w)
    cout = m.out_channels
    cdim = 1 if kind == L.CONVT_4X4_S2 else 0
    if cout % 8:
        pad = [0, 0] * (wp.dim() - 1 - cdim) + [0, _pad8(cout) - cout]
        wp = torch.nn.functional.pad(wp, pad)
    bias = torch.zeros(_pad8(cout), dtype=torch.float32, device=w.device)
    bias[:cout] = m.bias.detach().float()
    return kind, wp.contiguous(), bias, cout


def run_block_list(blocks, x, cache):
    """Interpret an nn.Sequential of the reference's block vocabulary (Conv2d 4x4s2 / 3x3 / 1x1, ConvTranspose2d 4x4s2 / 1x1,
    ReLU, ResBlock) on NHWC fp32 activations.  A ReLU that follows a convolution rides in that convolution's epilogue; a
    ReLU that follows a ResBlock (or opens one) is applied to the input of the next convolution inside the kernel.  Padded
    output channels (zero filters) are carried as zeros and ignored by the next layer's zero-padded input channels."""
    mods = list(blocks)
    relu_pending = False
    i = 0

    def conv(m, x, relu_out, relu_in, residual=None, relu_residual=False):
        key = (id(m), x.shape[-1])
        ver = (m.weight.data_ptr(), m.weight._version, m.bias._version)
        if key not in cache or cache[key][0] != ver:
            kind, wp, bias, _ = _pack_generic(m)
            if x.shape[-1] > wp.shape[-1]:               # the previous layer's zero-padded output channels: zero weights for them
                wp = torch.nn.functional.pad(wp, (0, x.shape[-1] - wp.shape[-1])).contiguous()
            cache[key] = (ver, (kind, wp, bias))
        kind, wp, bias = cache[key][1]
        if x.shape[-1] < wp.shape[-1]:
            x = torch.nn.functional.pad(x, (0, wp.shape[-1] - x.shape[-1]))
        return _conv(kind, x, wp, bias, bias.numel(), relu_out, relu_in=relu_in, residual=residual, relu_residual=relu_residual)

    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ReLU):
            relu_pending = True
            i += 1
        elif isinstance(m, ResBlock):
            c3, c1 = m.conv[1], m.conv[3]
            t = conv(c3, x, True, True)                  # leading ReLU on the input, second ReLU in the epilogue
            x = conv(c1, t, False, False, residual=x, relu_residual=True)      # + relu(input): the in-place ReLU's doing
            relu_pending = False                         # (a ReLU pending from before the block is subsumed by the block's own)
            i += 1
        else:
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = conv(m, x, fuse, relu_pending)
            relu_pending = False
            i += 2 if fuse else 1
    if relu_pending:
        x = torch.relu(x)
    return x


class Encoder(_ConvStack):
    """vqvae/vqvae_zc.py:116-164.  Production (stride 6, simple, no ResBlocks): conv4x4s2+ReLU, conv4x4s2+ReLU, conv4x4s2, ReLU,
    conv1x1 on the fused fast path; every other topology through run_block_list."""

    def __init__(self, in_channel, channel, n_res_block, n_res_channel, stride, embed_dim, n_embed, simple):
        if stride == 6:
            c1, c2 = (channel, channel) if simple else (channel // 4, channel // 2)
            blocks = [nn.Conv2d(in_channel, c1, 4, stride=2, padding=1), nn.ReLU(inplace=True),
                      nn.Conv2d(c1, c2, 4, stride=2, padding=1), nn.ReLU(inplace=True),
                      nn.Conv2d(c2, channel, 4, stride=2, padding=1)]
        elif stride == 4:
            blocks = [nn.Conv2d(in_channel, channel // 2, 4, stride=2, padding=1), nn.ReLU(inplace=True),
                      nn.Conv2d(channel // 2, channel, 4, stride=2, padding=1), nn.ReLU(inplace=True),
                      nn.Conv2d(channel, channel, 3, padding=1)]
        elif stride == 2:
            blocks = [nn.Conv2d(in_channel, channel // 2, 4, stride=2, padding=1), nn.ReLU(inplace=True),
                      nn.Conv2d(channel // 2, channel, 3, padding=1)]
        else:
            raise ValueError("stride must be 6, 4 or 2 (vqvae/vqvae_zc.py:120-155)")
        blocks += [ResBlock(channel, n_res_channel) for _ in range(n_res_block)]
        blocks += [nn.ReLU(inplace=True), nn.Conv2d(channel, embed_dim, 1)]
        super().__init__(blocks)
        self.channel, self.embed_dim = channel, embed_dim
        self._fast = stride == 6 and simple and n_res_block == 0 and channel % 8 == 0 and embed_dim % 8 == 0
        self._generic_cache = {}

    def forward(self, input):
        """NCHW image -> [b, h/8, w/8, embed_dim] (the reference returns the NHWC permutation too, :164)."""
        x = input.contiguous().float()
        b, c, h, w = x.shape
        assert c == 3, "encoder expects RGB input"
        x4 = torch.empty((b, h, w, 4), dtype=torch.float32, device=x.device)
        L.check(L.lib().cogv_nchw3_to_nhwc4_f32(_p(x), _p(x4), b, h, w, _stream()), "cogv_nchw3_to_nhwc4_f32")
        if not self._fast:
            y = run_block_list(self.blocks, x4, self._generic_cache)
            return y[..., :self.embed_dim].contiguous() if y.shape[-1] != self.embed_dim else y
        (w1, b1), (w2, b2), (w3, b3), (w4, b4) = self._weights([pack_conv_weight] * 4)
        y = _conv(L.CONV_4X4_S2, x4, w1, b1, self.channel, True)
        y = _conv(L.CONV_4X4_S2, y, w2, b2, self.channel, True)
        y = _conv(L.CONV_4X4_S2, y, w3, b3, self.channel, True)     # ReLU of blocks[5] folded into this epilogue
        return _conv(L.CONV_1X1, y, w4, b4, self.embed_dim, False)


class Decoder(_ConvStack):
    """vqvae/vqvae_zc.py:167-214.  Production (stride 4, simple, no ResBlocks): convT+ReLU x3, conv1x1 -> RGB on the fused fast
    path; every other topology through run_block_list."""

    def __init__(self, in_channel, out_channel, channel, n_res_block, n_res_channel, stride, simple):
        blocks = [nn.ConvTranspose2d(in_channel, channel, 4, stride=2, padding=1)]
        blocks += [ResBlock(channel, n_res_channel) for _ in range(n_res_block)]
        blocks.append(nn.ReLU(inplace=True))
        if stride == 4 and simple:
            blocks += [nn.ConvTranspose2d(channel, channel, 4, stride=2, padding=1), nn.ReLU(inplace=True),
                       nn.ConvTranspose2d(channel, channel, 4, stride=2, padding=1), nn.ReLU(inplace=True),
                       nn.Conv2d(channel, out_channel, 1)]
        elif stride == 4:
            blocks += [nn.ConvTranspose2d(channel, channel, 4, stride=2, padding=1), nn.ReLU(inplace=True),
                       nn.ConvTranspose2d(channel, channel // 2, 1), nn.ReLU(inplace=True),
                       nn.ConvTranspose2d(channel // 2, out_channel, 4, stride=2, padding=1)]
        elif stride == 2:
            blocks.append(nn.ConvTranspose2d(channel, out_channel, 4, stride=2, padding=1))
        # (any other stride: the reference appends nothing -- vqvae/vqvae_zc.py:178-208)
        super().__init__(blocks)
        # (a decoder stride other than 4 / 2 ends with the ReLU: its output has `channel` channels)
        self.channel, self.out_channel = channel, (out_channel if stride in (4, 2) else channel)
        self._fast = stride == 4 and simple and n_res_block == 0 and out_channel == 3 and channel % 8 == 0 and in_channel % 4 == 0
        self._generic_cache = {}

    def forward_nhwc(self, q_nhwc, scale=None, shift=None):
        if not self._fast:
            y = run_block_list(self.blocks, q_nhwc.contiguous().float(), self._generic_cache)
            img = y[..., :self.out_channel].permute(0, 3, 1, 2).contiguous()
            if scale is not None:
                img = img * torch.tensor(scale, device=img.device, dtype=img.dtype).view(1, -1, 1, 1)
            if shift is not None:
                img = img + torch.tensor(shift, device=img.device, dtype=img.dtype).view(1, -1, 1, 1)
            return img
        (w1, b1), (w2, b2), (w3, b3), (w4, b4) = self._weights([pack_convt_weight] * 3 + [lambda w: w.detach().float().reshape(3, -1).contiguous()])
        y = _conv(L.CONVT_4X4_S2, q_nhwc.contiguous(), w1, b1, self.channel, True)
        y = _conv(L.CONVT_4X4_S2, y, w2, b2, self.channel, True)
        if self.channel % 128 == 0:
            # production width: the last transposed convolution projects straight to RGB in its epilogue
            return _conv(L.CONVT_4X4_S2, y, w3, b3, self.channel, True, rgb=(w4, b4, scale, shift))
        y = _conv(L.CONVT_4X4_S2, y, w3, b3, self.channel, True)
        b, h, w, c = y.shape
        out = torch.empty((b, 3, h, w), dtype=torch.float32, device=y.device)
        sc = (C.c_float * 3)(*scale) if scale is not None else None
        sh = (C.c_float * 3)(*shift) if shift is not None else None
        with ops.timed_launch("conv1x1_rgb", 2.0 * b * h * w * c * 3, 4.0 * (y.numel() + out.numel()), f"{b}x{h}x{w}x{c}->3"):
            L.check(L.lib().cogv_conv1x1_to_rgb_f32(_p(y), _p(w4), _p(b4), _p(out), b, h, w, c, sc, sh, _stream()),
                    "cogv_conv1x1_to_rgb_f32")
        return out

    def forward(self, input):
        """NCHW quantised map -> NCHW image."""
        return self.forward_nhwc(input.permute(0, 2, 3, 1))


class VQVAE(nn.Module):
    def __init__(self, in_channel=3, channel=128, n_res_block=2, n_res_channel=32, embed_dim=64, n_embed=1024,
                 stride=4, simple=True, decay=0.99):
        super().__init__()
        if channel == 2048:
            n_res_block = 0
        self.enc_b = Encoder(in_channel, channel, n_res_block, n_res_channel, stride, embed_dim, n_embed, simple)
        self.quantize_t = Quantize(embed_dim, n_embed)
        self.dec = Decoder(in_channel=embed_dim, out_channel=in_channel, channel=channel, n_res_block=n_res_block,
                           n_res_channel=n_res_channel, stride=stride - 2, simple=simple)

    def encode(self, input, continuous_relax=False, temperature=1., hard=False, KL=False):
        """-> (quantised NCHW, diff [1], ids [b, h/8, w/8]) as vqvae/vqvae_zc.py:251-259."""
        logits = self.enc_b(input)
        quant_t, diff_t, id_t = self.quantize_t.forward_(logits, continuous_relax, temperature, hard)
        return quant_t.permute(0, 3, 1, 2), diff_t.unsqueeze(0), id_t

    def encode_ids(self, input):
        """ids only: skips the embedding lookup / diff that img2code discards."""
        return self.quantize_t.nearest_code(self.enc_b(input))

    def decode(self, code):
        return self.dec(code)

    def decode_code(self, code_t, scale=None, shift=None):
        return self.dec.forward_nhwc(self.quantize_t.embed_code(code_t), scale, shift)

    def forward(self, input, continuous_relax=False, temperature=1., hard=False, KL=False):
        quant_t, diff, _ = self.encode(input, continuous_relax, temperature, hard, KL)
        return self.dec(quant_t), diff
