"""Host-side mirror of the ToMP model predictor: `ltr/models/transformer/{transformer,filter_predictor,heads}.py`.

Same class names, constructor arguments, state_dict keys and call signatures as the reference, so a reference checkpoint
loads with `strict=True` and `pytracking/tracker/tomp/tomp.py:282-303` runs unchanged:

    Transformer(d_model, nhead, num_encoder_layers, num_decoder_layers, dim_feedforward, ...)     transformer.py:66-96
    FilterPredictor(transformer, feature_sz, use_test_frame_encoding)                             filter_predictor.py:20-150
        .predict_filter(train_feat, test_feat, train_label, train_ltrb_target)
        .predict_cls_bbreg_filters_parallel(train_feat, test_feat, train_label, num_gth_frames, train_ltrb_target)
    LinearFilterClassifier(num_channels, project_filter)                                          heads.py:83-98
    DenseBoxRegressor(num_channels, project_filter)                                               heads.py:101-141

The torch submodules (nn.MultiheadAttention, nn.Linear, ...) are parameter holders only: their forward is never called.
Inference only (the tracker runs under `torch.no_grad()`, tomp.py:284); training mode, pre-norm layers, non-ReLU
activations and multi-object attention raise NotImplementedError -- there is no stock-PyTorch fallback.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from . import filter as filter_layer
from .filter import _ptr, _require_device, _stream, device_guarded, workspace


class _Pack:
    """One contiguous fp32 device buffer holding a module's parameters in the order `include/pt_hot.h` documents;
    rebuilt when any source tensor was modified (optimizer step, load_state_dict) or moved."""

    def __init__(self):
        self.key = None
        self.buf = None

    def get(self, tensors):
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self.key:
            with torch.no_grad():
                self.buf = torch.cat([t.detach().reshape(-1).float() for t in tensors]).contiguous()
            self.key = key
        return self.buf


def _inference_only(mod):
    if mod.training:
        raise NotImplementedError(f"{type(mod).__name__}: the gfx950 path is inference only (call .eval())")


# ------------------------------------------------------------------------------------------------------------------
# transformer.py
# ------------------------------------------------------------------------------------------------------------------
class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if activation != "relu" or normalize_before:
            raise NotImplementedError("gfx950 ToMP path: post-norm layers with ReLU (the reference's configuration)")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def pack_list(self):
        a = self.self_attn
        return [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, self.linear1.weight,
                self.linear1.bias, self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias,
                self.norm2.weight, self.norm2.bias]


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        if activation != "relu" or normalize_before:
            raise NotImplementedError("gfx950 ToMP path: post-norm layers with ReLU (the reference's configuration)")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)

    def pack_list(self):
        out = []
        for a in (self.self_attn, self.multihead_attn):
            out += [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias]
        return out + [self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, self.norm1.weight,
                      self.norm1.bias, self.norm2.weight, self.norm2.bias, self.norm3.weight, self.norm3.bias]


class TransformerEncoder(nn.Module):
    def __init__(self, layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.num_layers = len(layers)
        self.norm = norm


class TransformerDecoder(nn.Module):
    def __init__(self, layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.num_layers = len(layers)
        self.norm = norm


class Transformer(nn.Module):
    """Parameter container with the reference's constructor (transformer.py:66-88); executed by FilterPredictor."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        if return_intermediate_dec:
            raise NotImplementedError("return_intermediate_dec is a training-time option")
        self.encoder = TransformerEncoder([TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation,
                                                                   normalize_before) for _ in range(num_encoder_layers)])
        self.decoder = TransformerDecoder([TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation,
                                                                   normalize_before) for _ in range(num_decoder_layers)],
                                          nn.LayerNorm(d_model))
        self._reset_parameters()
        self.d_model = d_model
        self.nhead = nhead
        self.dim_feedforward = dim_feedforward

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def pack_list(self):
        out = []
        for layer in self.encoder.layers:
            out += layer.pack_list()
        for layer in self.decoder.layers:
            out += layer.pack_list()
        return out + [self.decoder.norm.weight, self.decoder.norm.bias]

    def forward(self, *args, **kwargs):
        raise NotImplementedError("run through FilterPredictor (the fused predictor owns token build and both stacks)")


# ------------------------------------------------------------------------------------------------------------------
# filter_predictor.py
# ------------------------------------------------------------------------------------------------------------------
def MLP(channels, do_bn=True):
    layers = []
    for i in range(1, len(channels)):
        layers.append(nn.Conv1d(channels[i - 1], channels[i], kernel_size=1, bias=True))
        if i < len(channels) - 1:
            if do_bn:
                layers.append(nn.BatchNorm1d(channels[i]))
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class FilterPredictor(nn.Module):
    def __init__(self, transformer, feature_sz, use_test_frame_encoding=True):
        super().__init__()
        self.transformer = transformer
        self.feature_sz = feature_sz
        self.use_test_frame_encoding = use_test_frame_encoding
        d = self.transformer.d_model
        self.box_encoding = MLP([4, d // 4, d, d])
        self.query_embed_fg = nn.Embedding(1, d)
        if self.use_test_frame_encoding:
            self.query_embed_test = nn.Embedding(1, d)
        self.query_embed_fg_decoder = self.query_embed_fg
        self._pack = _Pack()
        self._pos = {}
        self._zero_tok = None
        self._prep_key, self._prepared = None, None

    def forward(self, train_feat, test_feat, train_label, train_ltrb_target, *args, **kwargs):
        return self.predict_filter(train_feat, test_feat, train_label, train_ltrb_target, *args, **kwargs)

    # ---- plumbing ---------------------------------------------------------------------------------------------
    def _params(self):
        be = self.box_encoding
        if self.use_test_frame_encoding:
            test_tok = self.query_embed_test.weight
        else:                                               # x + 0 == x exactly: the kernel always adds a test token
            if self._zero_tok is None or self._zero_tok.device != self.query_embed_fg.weight.device:
                self._zero_tok = torch.zeros_like(self.query_embed_fg.weight)
            test_tok = self._zero_tok
        tensors = self.transformer.pack_list() + [
            be[0].weight, be[0].bias, be[1].weight, be[1].bias, be[1].running_mean, be[1].running_var,
            be[3].weight, be[3].bias, be[4].weight, be[4].bias, be[4].running_mean, be[4].running_var,
            be[6].weight, be[6].bias, self.query_embed_fg.weight, test_tok]
        pack = self._pack.get(tensors)
        if self._prep_key != self._pack.key:                # weights changed: fold the decoder's weight products again
            L = _lib.lib()
            dims = self._dims(1, 1)
            self._prepared = torch.empty(L.pt_tomp_prepared_floats(ctypes.byref(dims)), dtype=torch.float32,
                                         device=pack.device)
            _lib.check(L.pt_tomp_prepare_f32(ctypes.byref(dims), _ptr(pack), _ptr(self._prepared), _stream()),
                       "pt_tomp_prepare_f32")
            self._prep_key = self._pack.key
        return pack, self._prepared

    def _dims(self, H, W):
        t = self.transformer
        fs = self.feature_sz
        fs = max(fs) if isinstance(fs, (list, tuple)) else fs
        return _lib.TompDims(t.d_model, t.nhead, t.dim_feedforward, len(t.encoder.layers), len(t.decoder.layers), H, W,
                             int(fs))

    @device_guarded
    def get_positional_encoding(self, feat):
        """(nframes, nseq, C, h, w) like the reference; the (h*w, C) table is computed once per map size."""
        nframes, nseq, C, h, w = feat.shape
        return self._pos_table(h, w, feat.device).t().reshape(1, 1, C, h, w).expand(nframes, nseq, C, h, w)

    def _pos_table(self, h, w, device):
        key = (h, w, device.index)
        if key not in self._pos:
            d = self._dims(h, w)
            pos = torch.empty(h * w, d.d_model, dtype=torch.float32, device=device)
            _lib.check(_lib.lib().pt_tomp_posenc_f32(_ptr(pos), h, w, d.d_model, d.max_res, _stream()),
                       "pt_tomp_posenc_f32")
            self._pos[key] = pos
        return self._pos[key]

    @device_guarded
    def _run(self, train_feat, test_feat, train_label, train_ltrb_target, parallel, num_gth_frames):
        _inference_only(self)
        if train_feat.dim() == 4:
            train_feat = train_feat.unsqueeze(1)
        if test_feat.dim() == 4:
            test_feat = test_feat.unsqueeze(1)
        if train_ltrb_target.dim() == 4:
            train_ltrb_target = train_ltrb_target.unsqueeze(1)
        _require_device(train_feat, test_feat, train_label, train_ltrb_target)
        nf, ns, D, H, W = train_feat.shape
        if test_feat.shape[0] != 1 or tuple(test_feat.shape[-2:]) != (H, W) or test_feat.shape[1] != ns:
            raise NotImplementedError("one test frame with the memory frames' map size (the tracker's configuration)")
        if D != self.transformer.d_model:
            raise ValueError("feature dimension does not match d_model")
        train_feat, test_feat = train_feat.contiguous(), test_feat.contiguous()
        train_label = train_label.reshape(nf, ns, H, W).contiguous()
        train_ltrb_target = train_ltrb_target.reshape(nf, ns, 4, H, W).contiguous()
        L = _lib.lib()
        dims = self._dims(H, W)
        nb = L.pt_tomp_predict_ws_bytes(ctypes.byref(dims), nf, ns, int(parallel))
        if nb == 0:
            raise NotImplementedError("FilterPredictor: configuration not covered by the gfx950 kernels")
        dev = train_feat.device
        ws = workspace(nb, dev)
        B = 2 if parallel else ns
        filters = torch.empty(B, D, dtype=torch.float32, device=dev)
        enc = torch.empty(B, D, H, W, dtype=torch.float32, device=dev)
        pack, prepared = self._params()
        rc = L.pt_tomp_predict_f32(ctypes.byref(dims), _ptr(pack), _ptr(prepared), _ptr(self._pos_table(H, W, dev)),
                                   _ptr(train_feat), _ptr(test_feat), _ptr(train_label), _ptr(train_ltrb_target), nf, ns,
                                   int(parallel), int(num_gth_frames), _ptr(filters), _ptr(enc), _ptr(ws), ws.numel(),
                                   _stream())
        _lib.check(rc, "pt_tomp_predict_f32")
        return filters, enc

    # ---- the reference's entry points ----------------------------------------------------------------------------
    def predict_filter(self, train_feat, test_feat, train_label, train_ltrb_target, *args, **kwargs):
        filters, enc = self._run(train_feat, test_feat, train_label, train_ltrb_target, False, 0)
        return filters.reshape(filters.shape[0], -1, 1, 1), enc.unsqueeze(0)

    def predict_cls_bbreg_filters_parallel(self, train_feat, test_feat, train_label, num_gth_frames, train_ltrb_target,
                                           *args, **kwargs):
        filters, enc = self._run(train_feat, test_feat, train_label, train_ltrb_target, True, num_gth_frames)
        D = filters.shape[1]
        return (filters[0].reshape(1, D, 1, 1), filters[1].reshape(1, D, 1, 1), enc[0][None, None], enc[1][None, None])


# ------------------------------------------------------------------------------------------------------------------
# heads.py
# ------------------------------------------------------------------------------------------------------------------
@device_guarded
def _project(linear, filt, C):
    x = filt.reshape(-1, C).contiguous()
    _require_device(x)
    if x.shape[0] > 8:
        raise NotImplementedError("more than 8 filters per call")
    y = torch.empty_like(x)
    rc = _lib.lib().pt_tomp_linear_f32(_ptr(linear.weight), _ptr(linear.bias), _ptr(x), _ptr(y), x.shape[0], C, C, 0,
                                       _stream())
    _lib.check(rc, "pt_tomp_linear_f32")
    return y.reshape(filt.shape)


class LinearFilterClassifier(nn.Module):
    def __init__(self, num_channels, project_filter=True):
        super().__init__()
        self.num_channels = num_channels
        self.project_filter = project_filter
        if project_filter:
            self.linear = nn.Linear(self.num_channels, self.num_channels)

    def forward(self, feat, filter):
        _inference_only(self)
        filter_proj = _project(self.linear, filter, self.num_channels) if self.project_filter else filter
        return filter_layer.apply_filter(feat, filter_proj)


def conv_layer(inplanes, outplanes, kernel_size=3, stride=1, padding=1, dilation=1):
    return [nn.Conv2d(inplanes, outplanes, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation),
            nn.GroupNorm(1, outplanes), nn.ReLU(inplace=True)]


class DenseBoxRegressor(nn.Module):
    def __init__(self, num_channels, project_filter=True):
        super().__init__()
        self.num_channels = num_channels
        self.project_filter = project_filter
        if self.project_filter:
            self.linear = nn.Linear(self.num_channels, self.num_channels)
        layers = []
        for _ in range(4):
            layers.extend(conv_layer(num_channels, num_channels))
        self.tower = nn.Sequential(*layers)
        self.bbreg_layer = nn.Conv2d(num_channels, 4, kernel_size=3, dilation=1, padding=1)
        self._pack = _Pack()
        self._eye = None

    def _params(self):
        C = self.num_channels
        if self.project_filter:
            lw, lb = self.linear.weight, self.linear.bias
        else:                                               # identity projection through the same kernel
            if self._eye is None or self._eye[0].device != self.bbreg_layer.weight.device:
                dev = self.bbreg_layer.weight.device
                self._eye = (torch.eye(C, device=dev), torch.zeros(C, device=dev))
            lw, lb = self._eye
        tensors = [lw, lb]
        for i in range(4):
            conv, gn = self.tower[3 * i], self.tower[3 * i + 1]
            tensors += [conv.weight.permute(0, 2, 3, 1), conv.bias, gn.weight, gn.bias]     # (out, ky, kx, in)
        tensors += [self.bbreg_layer.weight.permute(0, 2, 3, 1), self.bbreg_layer.bias]
        # permute() returns views sharing the parameters' version counters, so the cache key still tracks updates
        return self._pack.get(tensors)

    @device_guarded
    def forward(self, feat, filter):
        _inference_only(self)
        nf, ns, c, h, w = feat.shape
        _require_device(feat, filter)
        if filter.numel() != ns * c:
            raise NotImplementedError("one filter per sequence (multi-object attention is not on the hot path)")
        L = _lib.lib()
        params = self._params()
        out = torch.empty(nf, ns, 4, h, w, dtype=torch.float32, device=feat.device)
        nb = L.pt_tomp_bbreg_ws_bytes(nf, c, h, w)
        if nb == 0:
            raise NotImplementedError("DenseBoxRegressor: configuration not covered by the gfx950 kernels")
        ws = workspace(nb, feat.device)
        filt = filter.reshape(ns, c).contiguous()
        for s in range(ns):
            fs = feat[:, s].contiguous()
            os_ = out[:, s] if ns == 1 else torch.empty(nf, 4, h, w, dtype=torch.float32, device=feat.device)
            rc = L.pt_tomp_bbreg_f32(_ptr(params), _ptr(fs), _ptr(filt[s]), _ptr(os_), nf, c, h, w, _ptr(ws), ws.numel(),
                                     _stream())
            _lib.check(rc, "pt_tomp_bbreg_f32")
            if ns != 1:
                out[:, s] = os_
        return out.reshape(1, nf * ns, 4, h, w)
