"""GPU test: the ctypes stubs printed in INTEGRATION.md (what a maintainer of the reference would add) drive the library correctly.
The stub code below is the INTEGRATION.md text; it uses nothing from the mi355attn Python package except the path of the built .so."""
import ctypes

import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


@pytest.fixture(scope="module")
def _lib():
    import mi355attn
    lib = ctypes.CDLL(mi355attn.LIB_PATH)
    lib.mi355_se_workspace_bytes.restype = _sz
    lib.mi355_se_workspace_bytes.argtypes = [_i] * 4
    lib.mi355_se_fwd.restype = _i
    lib.mi355_se_fwd.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]
    lib.mi355_bam_workspace_bytes.restype = _sz
    lib.mi355_bam_workspace_bytes.argtypes = [_i] * 5
    lib.mi355_bam_fwd.restype = _i
    lib.mi355_bam_fwd.argtypes = [_vp, _vp, _vp] + [_i] * 6 + [_vp, _sz, _vp]
    lib.mi355_cast16_fwd.restype = _i
    lib.mi355_cast16_fwd.argtypes = [_vp, _vp, _sz, _i, _vp]
    lib.mi355_mhsa_workspace_bytes.restype = _sz
    lib.mi355_mhsa_workspace_bytes.argtypes = [_i] * 4
    lib.mi355_mhsa_fwd.restype = _i
    lib.mi355_mhsa_fwd.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _i, _vp, _sz, _vp]
    lib.mi355_last_error.restype = ctypes.c_char_p
    return lib


def test_se_stub(_lib):
    torch.manual_seed(0)
    x = torch.randn(3, 64, 20, 28).cuda()
    w1, w2 = (torch.randn(4, 64) / 8).cuda(), (torch.randn(64, 4) / 2).cuda()
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    ws = torch.empty(_lib.mi355_se_workspace_bytes(B, C, H, W), dtype=torch.uint8, device=x.device)
    rc = _lib.mi355_se_fwd(x.data_ptr(), w1.data_ptr(), w2.data_ptr(), y.data_ptr(), B, C, w1.shape[0], H, W, ws.data_ptr(), ws.numel(),
                           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.mi355_last_error().decode()
    ref = O.se_forward(x.cpu(), w1.cpu(), w2.cpu())
    assert float((y.cpu() - ref).norm() / ref.norm()) < 1e-5


def _fold(bn, pre_bias=None):
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    t = bn.bias - bn.running_mean * s + (s * pre_bias if pre_bias is not None else 0)
    return s.contiguous(), t.contiguous()


def test_bam_stub(_lib):
    from mi355attn.modules import BAM                       # same submodule names / parameters as the reference's BAM
    torch.manual_seed(1)
    m = BAM(64).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.2 * torch.randn_like(p))
        for mod in m.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 64, 16, 20)
    ref = O.bam_forward(x, m.state_dict())
    m = m.cuda(); x = x.cuda()
    with torch.no_grad():
        ch, sp = m.channel_attn, m.spatial_attn
        bn1d, d1, d2 = _fold(ch.bn), _fold(sp.conv2[1], sp.conv2[0].bias), _fold(sp.conv2[4], sp.conv2[3].bias)
        s3, t3 = _fold(sp.bn, sp.conv3.bias)
        ps = [ch.mlp[0].weight, ch.mlp[0].bias, ch.mlp[2].weight, ch.mlp[2].bias, *bn1d, sp.conv1.weight, sp.conv1.bias,
              sp.conv2[0].weight, *d1, sp.conv2[3].weight, *d2, (sp.conv3.weight.reshape(-1) * s3).contiguous(), t3]
        ps = [p.detach().contiguous() for p in ps]
        table = (ctypes.c_void_p * 16)(*[p.data_ptr() for p in ps])
        B, C, H, W = x.shape
        Cr = sp.conv1.weight.shape[0]
        y = torch.empty_like(x)
        ws = torch.empty(_lib.mi355_bam_workspace_bytes(B, C, Cr, H, W), dtype=torch.uint8, device=x.device)
        rc = _lib.mi355_bam_fwd(x.data_ptr(), ctypes.cast(table, _vp), y.data_ptr(), B, C, Cr, H, W, 4, ws.data_ptr(), ws.numel(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.mi355_last_error().decode()
    assert float((y.cpu() - ref).norm() / ref.norm()) < 3e-5


def test_mhsa_stub(_lib):
    """Attention.forward (ViT.py:79-89) through mi355_mhsa_fwd with nothing but ctypes."""
    from mi355attn.modules import Attention                 # parameter container with the reference's names
    torch.manual_seed(1234)
    m = Attention(768, 12, qkv_bias=True).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(3, 197, 768)
    ref = O.vit_attention_forward(x, sd, 12)
    m = m.cuda(); x = x.cuda()
    st = torch.cuda.current_stream().cuda_stream

    def w16(w):
        out = torch.empty(w.shape, dtype=torch.float16, device=w.device)
        assert _lib.mi355_cast16_fwd(w.detach().contiguous().data_ptr(), out.data_ptr(), w.numel(), 1, st) == 0
        return out

    B, N, C = x.shape
    wqkv, wproj = w16(m.qkv.weight), w16(m.proj.weight)
    y = torch.empty_like(x)
    nws = _lib.mi355_mhsa_workspace_bytes(B, N, C, 0)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
    rc = _lib.mi355_mhsa_fwd(x.data_ptr(), 0, wqkv.data_ptr(), m.qkv.bias.data_ptr(), wproj.data_ptr(), m.proj.bias.data_ptr(), None,
                             y.data_ptr(), B, N, C, 12, float(m.scale), 1, ws.data_ptr(), nws, st)
    assert rc == 0, _lib.mi355_last_error().decode()
    err = float((y.cpu() - ref).norm() / ref.norm())
    assert err < 1e-3, err
    # too small a workspace is refused before anything is launched
    rc = _lib.mi355_mhsa_fwd(x.data_ptr(), 0, wqkv.data_ptr(), None, wproj.data_ptr(), None, None, y.data_ptr(), B, N, C, 12, float(m.scale), 1,
                             ws.data_ptr(), 1024, st)
    assert rc == -1 and b"invalid argument" in _lib.mi355_last_error()
