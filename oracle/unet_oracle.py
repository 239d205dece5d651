"""
ORACLE  --  TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (numpy) restatement of the reference's train/evaluate hot path:
U-Net / MulmoU-Net forward, weighted-BCE loss, backward and Keras-Adam.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the reported CPU baseline.

PARITY UNPINNED by the reference: /root/reference holds no golden vectors, no
known-answer tests and no fixtures for this path (its only tests cover region
metrics), and its arithmetic lives in TensorFlow 2.6.0 + tensorflow-addons
(requirements.txt:2-3), which are not vendored and not installed here.  This
restatement follows the reference's call sites and the published TF-2.6 / Keras
semantics of the ops they call, and is pinned instead by an independent
torch-CPU autograd cross-check (tests/golden/make_golden.py, tests/test_oracle.py).

Reference lines followed (all under /root/reference/annotator/):
  models/tf_models/components.py:16-81    Downsample  (Conv2D+act [+BN]) x n_conv -> skip; MaxPool [+BN]
  models/tf_models/components.py:84-166   Upsample    ConvT(k=s=rate) [+BN]; crop; concat([tconv, skip]); convs
  models/tf_models/components.py:169-247  Encoder     filters f0 * rate^i
  models/tf_models/components.py:250-320  Decoder     one Upsample per skip, deepest first, filters = skip channels
  models/tf_models/unet.py:19-88          UNet        (no bottleneck convs)
  models/tf_models/unet.py:91-191         MulmoUNet   one encoder per input channel, bottleneck concat, skips of encoder[reference_index]
  models/tf_models/unet.py:194-300        UNetAnnotator / MulmoUNetAnnotator: Conv2D(1, 1, activation='sigmoid') head
  utils/losses.py:17-37, 87-102           weighted cross-entropy from logits, positive-rate weight
  engine.py:276-284                       Adam(lr=1e-3, b1=.9, b2=.999, eps=1e-7)
"""

from collections import OrderedDict

import numpy as np


# --------------------------------------------------------------------------------------
# model description
# --------------------------------------------------------------------------------------

class ModelSpec:
    """Hyper-parameters of one annotator model (unet.py:195-207 signature)."""

    def __init__(self, arch, in_channels, n_filters_first, n_downsample, rate=2, kernel_size=3,
                 conv_stride=1, bn=False, padding='valid', activation='relu', kernel_regularizer=None,
                 reference_index=0, n_conv=2):
        assert arch in ('unet', 'mulmo')
        assert conv_stride == 1, 'the reference configs only use conv_stride 1'
        assert padding in ('same', 'valid')
        self.arch = arch
        self.in_channels = int(in_channels)
        self.f0 = int(n_filters_first)
        self.n_down = int(n_downsample)
        self.rate = int(rate)
        self.k = int(kernel_size)
        self.bn = bool(bn)
        self.padding = padding
        self.n_conv = n_conv
        self.reference_index = reference_index
        # activation: 'relu' or {'class_name': 'LeakyReLU', 'config': {'alpha': a}} (components.py:323-335)
        if isinstance(activation, str):
            assert activation == 'relu', activation
            self.alpha = 0.0
        else:
            assert activation['class_name'] == 'LeakyReLU'
            self.alpha = float(activation.get('config', {}).get('alpha', 0.3))
        # kernel_regularizer: None or {'class_name': 'L2', 'config': {'l2': x}} (kernel_regularizer.yaml:1-4)
        if kernel_regularizer is None:
            self.l2 = 0.0
        else:
            assert kernel_regularizer['class_name'] == 'L2'
            self.l2 = float(kernel_regularizer.get('config', {}).get('l2', 0.01))

    @classmethod
    def from_config(cls, config, in_channels):
        """config = the reference's YAML dict (unet.yaml:1-9): {'model': name, 'model_options': {...}}."""
        arch = {'UNetAnnotator': 'unet', 'MulmoUNetAnnotator': 'mulmo'}[config['model']]
        return cls(arch=arch, in_channels=in_channels, **config['model_options'])

    # ---- structure -------------------------------------------------------------------
    def encoder_filters(self):
        f, out = self.f0, []
        for _ in range(self.n_down):           # components.py:204-220
            out.append(f)
            f = int(self.rate * f)
        return out

    def n_encoders(self):
        return self.in_channels if self.arch == 'mulmo' else 1

    def encoder_in_channels(self):
        return 1 if self.arch == 'mulmo' else self.in_channels


def _bn_specs(prefix, c):
    return [(prefix + '.gamma', (c,), True), (prefix + '.beta', (c,), True),
            (prefix + '.moving_mean', (c,), False), (prefix + '.moving_variance', (c,), False)]


def param_specs(spec):
    """[(name, shape, trainable)] in Keras variable-creation order:
    Encoder.build -> Downsample.build (convchain, then pool) components.py:69-75,226-233;
    Decoder.build components.py:292-312; head unet.py:273-277; Mulmo builds encoders first unet.py:152-176."""
    out = []
    k = spec.k
    filters = spec.encoder_filters()
    for e in range(spec.n_encoders()):
        enc = 'encoder%d' % e if spec.arch == 'mulmo' else 'encoder'
        cin = spec.encoder_in_channels()
        for i, f in enumerate(filters):
            c = cin
            for j in range(spec.n_conv):
                p = '%s.down%d' % (enc, i)
                out += [(p + '.conv%d.kernel' % j, (k, k, c, f), True), (p + '.conv%d.bias' % j, (f,), True)]
                if spec.bn:
                    out += _bn_specs(p + '.bn%d' % j, f)
                c = f
            if spec.bn:
                out += _bn_specs('%s.down%d.pool_bn' % (enc, i), f)
            cin = f
    cin = filters[-1] * spec.n_encoders()
    r = spec.rate
    for u, f in enumerate(reversed(filters)):
        p = 'decoder.up%d' % u
        # Conv2DTranspose kernel layout [kh, kw, Cout, Cin]
        out += [(p + '.tconv.kernel', (r, r, f, cin), True), (p + '.tconv.bias', (f,), True)]
        if spec.bn:
            out += _bn_specs(p + '.tconv_bn', f)
        c = 2 * f
        for j in range(spec.n_conv):
            out += [(p + '.conv%d.kernel' % j, (k, k, c, f), True), (p + '.conv%d.bias' % j, (f,), True)]
            if spec.bn:
                out += _bn_specs(p + '.bn%d' % j, f)
            c = f
        cin = f
    out += [('head.kernel', (1, 1, filters[0], 1), True), ('head.bias', (1,), True)]
    return out


def init_params(spec, seed=2, dtype=np.float32):
    """Keras defaults: glorot_uniform kernels, zero biases, BN gamma 1 / beta 0 / mean 0 / var 1."""
    rng = np.random.default_rng(seed)
    params = OrderedDict()
    for name, shape, _ in param_specs(spec):
        if name.endswith('.kernel'):
            kh, kw, a, b = shape
            # fan_in + fan_out = kh*kw*(Cin + Cout) for Conv2D [kh,kw,Cin,Cout] and ConvT [kh,kw,Cout,Cin] alike
            limit = np.sqrt(6.0 / (kh * kw * (a + b)))
            params[name] = rng.uniform(-limit, limit, size=shape).astype(dtype)
        elif name.endswith('.gamma') or name.endswith('.moving_variance'):
            params[name] = np.ones(shape, dtype)
        else:
            params[name] = np.zeros(shape, dtype)
    return params


def flatten(spec, params, trainable=True):
    return np.concatenate([np.asarray(params[n]).ravel() for n, _, t in param_specs(spec) if t == trainable]
                          or [np.zeros(0, np.float32)])


def unflatten(spec, flat, trainable=True, into=None):
    out = OrderedDict() if into is None else into
    off = 0
    for n, shape, t in param_specs(spec):
        if t != trainable:
            continue
        size = int(np.prod(shape))
        out[n] = np.asarray(flat[off:off + size]).reshape(shape).copy()
        off += size
    assert off == len(flat), (off, len(flat))
    return out


# --------------------------------------------------------------------------------------
# layers (each fwd returns (out, cache); each bwd returns (d_in, {grad name: array}))
# --------------------------------------------------------------------------------------

def _act_fwd(z, alpha):
    return np.where(z > 0, z, alpha * z) if alpha != 0.0 else np.maximum(z, 0)


def _act_bwd(y, dy, alpha):
    # gradient is dy * [y > 0] (+ alpha * [y <= 0]); sign(y) == sign(z) for alpha >= 0
    return dy * np.where(y > 0, 1.0, alpha).astype(dy.dtype)


def conv2d_fwd(x, w, b, padding, alpha=None):
    """Keras Conv2D, stride 1: cross-correlation, kernel HWIO, zero padding for 'same' (components.py:46-52)."""
    kh, kw, cin, cout = w.shape
    B, H, W, C = x.shape
    assert C == cin, (x.shape, w.shape)
    if padding == 'same':
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        xp = np.pad(x, ((0, 0), (ph, kh - 1 - ph), (pw, kw - 1 - pw), (0, 0)))
        Ho, Wo = H, W
    else:
        xp = x
        Ho, Wo = H - kh + 1, W - kw + 1
    z = np.zeros((B, Ho, Wo, cout), x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            z += xp[:, ky:ky + Ho, kx:kx + Wo, :] @ w[ky, kx]
    z += b
    y = z if alpha is None else _act_fwd(z, alpha)
    return y, (xp, w, y, alpha, padding, x.shape)


def conv2d_bwd(cache, dy):
    xp, w, y, alpha, padding, xshape = cache
    kh, kw, cin, cout = w.shape
    B, Ho, Wo, _ = dy.shape
    dz = dy if alpha is None else _act_bwd(y, dy, alpha)
    dw = np.zeros_like(w)
    dxp = np.zeros_like(xp)
    dz2 = dz.reshape(-1, cout)
    for ky in range(kh):
        for kx in range(kw):
            xs = xp[:, ky:ky + Ho, kx:kx + Wo, :]
            dw[ky, kx] = xs.reshape(-1, cin).T @ dz2
            dxp[:, ky:ky + Ho, kx:kx + Wo, :] += dz @ w[ky, kx].T
    db = dz2.sum(0)
    if padding == 'same':
        ph, pw = (kh - 1) // 2, (kw - 1) // 2
        H, W = xshape[1], xshape[2]
        dx = dxp[:, ph:ph + H, pw:pw + W, :]
    else:
        dx = dxp
    return dx, dw, db


def maxpool_fwd(x, r):
    """MaxPool2D([r, r], strides=r), VALID (components.py:54)."""
    B, H, W, C = x.shape
    Ho, Wo = H // r, W // r
    xw = x[:, :Ho * r, :Wo * r, :].reshape(B, Ho, r, Wo, r, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, Ho, Wo, C, r * r)
    idx = xw.argmax(-1)                  # first maximum in row-major window order
    y = np.take_along_axis(xw, idx[..., None], -1)[..., 0]
    return y, (idx, x.shape, r)


def maxpool_bwd(cache, dy):
    idx, xshape, r = cache
    B, H, W, C = xshape
    Ho, Wo = H // r, W // r
    dxw = np.zeros((B, Ho, Wo, C, r * r), dy.dtype)
    np.put_along_axis(dxw, idx[..., None], dy[..., None], -1)
    dx = np.zeros(xshape, dy.dtype)
    dx[:, :Ho * r, :Wo * r, :] = dxw.reshape(B, Ho, Wo, C, r, r).transpose(0, 1, 4, 2, 5, 3).reshape(B, Ho * r, Wo * r, C)
    return dx


def tconv_fwd(x, w, b):
    """Conv2DTranspose(k=r, s=r, activation=None) (components.py:118-120); kernel [kh,kw,Cout,Cin];
    out[b, r*i+a, r*j+c, co] = sum_ci x[b,i,j,ci] * w[a,c,co,ci] + bias[co]."""
    r = w.shape[0]
    B, H, W, cin = x.shape
    cout = w.shape[2]
    y = np.empty((B, H, r, W, r, cout), x.dtype)
    for a in range(r):
        for c in range(r):
            y[:, :, a, :, c, :] = x @ w[a, c].T + b
    return y.reshape(B, H * r, W * r, cout), (x, w)


def tconv_bwd(cache, dy):
    x, w = cache
    r = w.shape[0]
    B, H, W, cin = x.shape
    cout = w.shape[2]
    dyr = dy.reshape(B, H, r, W, r, cout)
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    x2 = x.reshape(-1, cin)
    for a in range(r):
        for c in range(r):
            g = dyr[:, :, a, :, c, :]
            dx += g @ w[a, c]
            dw[a, c] = g.reshape(-1, cout).T @ x2
    db = dy.reshape(-1, cout).sum(0)
    return dx, dw, db


BN_EPS = 1e-3        # Keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99   # Keras default momentum


def bn_fwd(x, gamma, beta, mmean, mvar, training):
    """Keras BatchNormalization(axis=-1) [TF-2.6 fused semantics]: train = biased batch variance for
    normalisation, moving_var updated with the unbiased one; inference = moving statistics."""
    if training:
        n = x.shape[0] * x.shape[1] * x.shape[2]
        mean = x.mean((0, 1, 2), dtype=np.float64).astype(x.dtype) if x.dtype == np.float32 else x.mean((0, 1, 2))
        xc = x - mean
        var = (xc * xc).mean((0, 1, 2), dtype=np.float64).astype(x.dtype) if x.dtype == np.float32 \
            else (xc * xc).mean((0, 1, 2))
        inv = 1.0 / np.sqrt(var + x.dtype.type(BN_EPS))
        xhat = xc * inv
        y = xhat * gamma + beta
        unbiased = var * (n / max(n - 1, 1))
        new_mean = mmean * BN_MOMENTUM + mean * (1 - BN_MOMENTUM)
        new_var = mvar * BN_MOMENTUM + unbiased * (1 - BN_MOMENTUM)
        return y, (xhat, inv, gamma, True), (new_mean.astype(x.dtype), new_var.astype(x.dtype))
    inv = 1.0 / np.sqrt(mvar + x.dtype.type(BN_EPS))
    y = (x - mmean) * (inv * gamma) + beta
    return y, ((x - mmean) * inv, inv, gamma, False), (mmean, mvar)


def bn_bwd(cache, dy):
    xhat, inv, gamma, training = cache
    dgamma = (dy * xhat).sum((0, 1, 2))
    dbeta = dy.sum((0, 1, 2))
    if training:
        n = dy.shape[0] * dy.shape[1] * dy.shape[2]
        dx = (gamma * inv / n) * (n * dy - dbeta - xhat * dgamma)
    else:
        dx = dy * (gamma * inv)
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------------------
# network
# --------------------------------------------------------------------------------------

class Tape:
    """records (bwd closure) in forward order; run() replays them in reverse."""

    def __init__(self):
        self.grads = OrderedDict()

    def add(self, name, g):
        self.grads[name] = self.grads[name] + g if name in self.grads else g


def _crop_center(ref, h, w):
    gh, gw = (ref.shape[1] - h) // 2, (ref.shape[2] - w) // 2     # components.py:162-163
    return ref[:, gh:gh + h, gw:gw + w, :], (gh, gw)


def forward(spec, params, x, training=False, keep=False):
    """Returns (logits [B,H',W',1], new_state dict, backward closure or None, activations dict if keep).

    Order inside a block is Conv -> bias -> activation -> BN (components.py:56-59,129-132)."""
    P = params
    new_state = {}
    acts = OrderedDict()
    bw = []       # list of closures taking/returning upstream gradient(s)

    def bn(prefix, t):
        y, cache, (nm, nv) = bn_fwd(t, P[prefix + '.gamma'], P[prefix + '.beta'],
                                    P[prefix + '.moving_mean'], P[prefix + '.moving_variance'], training)
        new_state[prefix + '.moving_mean'] = nm
        new_state[prefix + '.moving_variance'] = nv
        return y, cache

    def down(prefix, t):
        caches = []
        for j in range(spec.n_conv):
            t, cc = conv2d_fwd(t, P['%s.conv%d.kernel' % (prefix, j)], P['%s.conv%d.bias' % (prefix, j)],
                               spec.padding, spec.alpha)
            bc = None
            if spec.bn:
                t, bc = bn('%s.bn%d' % (prefix, j), t)
            caches.append((cc, bc))
        skip = t
        half, pc = maxpool_fwd(skip, spec.rate)
        pbc = None
        if spec.bn:
            half, pbc = bn(prefix + '.pool_bn', half)
        return skip, half, (caches, pc, pbc)

    def down_bwd(prefix, cache, dskip, dhalf, tape):
        caches, pc, pbc = cache
        if pbc is not None:
            dhalf, dg, db = bn_bwd(pbc, dhalf)
            tape.add(prefix + '.pool_bn.gamma', dg)
            tape.add(prefix + '.pool_bn.beta', db)
        d = maxpool_bwd(pc, dhalf)
        if dskip is not None:
            d = d + dskip
        for j in reversed(range(spec.n_conv)):
            cc, bc = caches[j]
            if bc is not None:
                d, dg, db = bn_bwd(bc, d)
                tape.add('%s.bn%d.gamma' % (prefix, j), dg)
                tape.add('%s.bn%d.beta' % (prefix, j), db)
            d, dw, dbias = conv2d_bwd(cc, d)
            tape.add('%s.conv%d.kernel' % (prefix, j), dw)
            tape.add('%s.conv%d.bias' % (prefix, j), dbias)
        return d

    filters = spec.encoder_filters()
    enc_caches, skips_all, bottoms = [], [], []
    for e in range(spec.n_encoders()):
        enc = 'encoder%d' % e if spec.arch == 'mulmo' else 'encoder'
        t = x[..., e:e + 1] if spec.arch == 'mulmo' else x        # unet.py:183
        caches, skips = [], []
        for i in range(spec.n_down):
            skip, t, c = down('%s.down%d' % (enc, i), t)
            skips.append(skip)
            caches.append(c)
            if keep:
                acts['%s.down%d.skip' % (enc, i)] = skip
                acts['%s.down%d.half' % (enc, i)] = t
        enc_caches.append(caches)
        skips_all.append(skips)
        bottoms.append(t)
    t = np.concatenate(bottoms, -1) if spec.arch == 'mulmo' else bottoms[0]     # unet.py:187
    ref = skips_all[spec.reference_index if spec.arch == 'mulmo' else 0]        # unet.py:188

    up_caches = []
    for u in range(spec.n_down):
        p = 'decoder.up%d' % u
        reference = ref[spec.n_down - 1 - u]
        tc, tcache = tconv_fwd(t, P[p + '.tconv.kernel'], P[p + '.tconv.bias'])
        tbc = None
        if spec.bn:
            tc, tbc = bn(p + '.tconv_bn', tc)
        cropped, off = _crop_center(reference, tc.shape[1], tc.shape[2])
        t = np.concatenate([tc, cropped], -1)                      # components.py:164 (up-sampled first)
        caches = []
        for j in range(spec.n_conv):
            t, cc = conv2d_fwd(t, P['%s.conv%d.kernel' % (p, j)], P['%s.conv%d.bias' % (p, j)], spec.padding, spec.alpha)
            bc = None
            if spec.bn:
                t, bc = bn('%s.bn%d' % (p, j), t)
            caches.append((cc, bc))
        up_caches.append((tcache, tbc, off, reference.shape, tc.shape[-1], caches))
        if keep:
            acts[p + '.tconv'] = tc
            acts[p + '.out'] = t
    feat = t
    logits, hcache = conv2d_fwd(feat, P['head.kernel'], P['head.bias'], spec.padding, None)   # unet.py:241-244
    if keep:
        acts['logits'] = logits

    def backward(dlogits):
        tape = Tape()
        d, dw, db = conv2d_bwd(hcache, dlogits)
        tape.add('head.kernel', dw)
        tape.add('head.bias', db)
        dskips = [None] * spec.n_down
        for u in reversed(range(spec.n_down)):
            p = 'decoder.up%d' % u
            tcache, tbc, off, refshape, ctc, caches = up_caches[u]
            for j in reversed(range(spec.n_conv)):
                cc, bc = caches[j]
                if bc is not None:
                    d, dg, dbt = bn_bwd(bc, d)
                    tape.add('%s.bn%d.gamma' % (p, j), dg)
                    tape.add('%s.bn%d.beta' % (p, j), dbt)
                d, dw, db = conv2d_bwd(cc, d)
                tape.add('%s.conv%d.kernel' % (p, j), dw)
                tape.add('%s.conv%d.bias' % (p, j), db)
            dtc, dcrop = d[..., :ctc], d[..., ctc:]
            dref = np.zeros(refshape, d.dtype)
            dref[:, off[0]:off[0] + d.shape[1], off[1]:off[1] + d.shape[2], :] = dcrop
            dskips[spec.n_down - 1 - u] = dref
            if tbc is not None:
                dtc, dg, dbt = bn_bwd(tbc, dtc)
                tape.add(p + '.tconv_bn.gamma', dg)
                tape.add(p + '.tconv_bn.beta', dbt)
            d, dw, db = tconv_bwd(tcache, dtc)
            tape.add(p + '.tconv.kernel', dw)
            tape.add(p + '.tconv.bias', db)
        dx_parts = []
        cb = filters[-1]
        for e in range(spec.n_encoders()):
            enc = 'encoder%d' % e if spec.arch == 'mulmo' else 'encoder'
            de = d[..., e * cb:(e + 1) * cb] if spec.arch == 'mulmo' else d
            is_ref = (e == (spec.reference_index if spec.arch == 'mulmo' else 0))
            for i in reversed(range(spec.n_down)):
                de = down_bwd('%s.down%d' % (enc, i), enc_caches[e][i], dskips[i] if is_ref else None, de, tape)
            dx_parts.append(de)
        dx = np.concatenate(dx_parts, -1) if spec.arch == 'mulmo' else dx_parts[0]
        return dx, tape.grads

    return logits, new_state, backward, acts


# --------------------------------------------------------------------------------------
# loss (utils/losses.py)
# --------------------------------------------------------------------------------------

def positive_rate(label):
    """utils/losses.py:87-102: sum(label) / numel(label); asserts 0 <= label <= 1."""
    if label.size:
        assert label.max() <= 1.0 and label.min() >= 0.0, 'label out of [0, 1]'
    return label.sum(dtype=np.float64) / max(label.size, 1)


def loss_weight(label, weight=None, weight_add=0.0, weight_mul=1.0):
    """utils/losses.py:25-30."""
    if weight is None:
        pr = positive_rate(label)
        weight = 1.0 / pr if pr > 0.0 else 1.0
    weight = weight_mul * weight + weight_add
    assert weight >= 0.0, 'assert_on_weight'
    return weight


def gaussian_filter2d(label, filter_size=6, sigma=3.0):
    """tfa.image.gaussian_filter2d(label[..., None], filter_shape, sigma)[..., 0] as called at utils/losses.py:64-66
    (tensorflow-addons: third-party, absent, unpinned in requirements.txt:3; restated from its published algorithm):
    1-D kernel softmax(-u^2 / (2 sigma^2)) over u = range(-k // 2 + 1, k // 2 + 1), 2-D kernel = outer product, REFLECT
    padding of (k - 1) // 2 before and k - 1 - (k - 1) // 2 after, VALID correlation.  float64 arithmetic."""
    k = int(filter_size)
    u = np.arange(-k // 2 + 1, k // 2 + 1, dtype=np.float64)
    g = np.exp(-(u ** 2) / (2.0 * float(sigma) ** 2))
    g /= g.sum()
    before = (k - 1) // 2
    after = k - 1 - before
    pad = np.pad(np.asarray(label, np.float64), ((0, 0), (before, after), (before, after)), mode='reflect')
    H, W = label.shape[1:]
    out = np.zeros(label.shape, np.float64)
    for i in range(k):
        for j in range(k):
            out += g[i] * g[j] * pad[:, i:i + H, j:j + W]
    return out


def weighted_crossentropy(label, logits, weight=None, weight_add=0.0, weight_mul=1.0, label_smoothing=False,
                          label_smoothing_filter_size=6, label_smoothing_sigma=3):
    """utils/losses.py:17-37 with from_logits=True: returns (per-sample loss [B], dloss_b/dlogits [B,H,W,1]).

    BCE-with-logits [TF semantics]: max(x,0) - x*z + log1p(exp(-|x|)); the BCE's own mean over the size-1
    channel axis is a no-op; sample_weight = label*(w-1)+1; mean over (H, W).  label_smoothing (TFWeightedCrossentropy.call,
    utils/losses.py:62-67) blurs the labels first; the positive rate is then taken from the blurred labels."""
    dt = logits.dtype
    if label.shape[0] == 0:
        return np.zeros([0], dt), np.zeros_like(logits)
    if label_smoothing:
        label = gaussian_filter2d(label, label_smoothing_filter_size, label_smoothing_sigma).astype(label.dtype)
    w = loss_weight(label, weight, weight_add, weight_mul)
    z = label.astype(dt)
    mask = z * dt.type(w - 1.0) + dt.type(1.0)
    x = logits[..., 0]
    bce = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    per = (bce * mask).mean((1, 2))
    sig = 1.0 / (1.0 + np.exp(-x))
    dper = (mask * (sig - z) / dt.type(x.shape[1] * x.shape[2]))[..., None]
    return per.astype(dt), dper.astype(dt)


def l2_penalty(spec, params):
    if spec.l2 == 0.0:
        return 0.0
    return spec.l2 * sum(float((np.asarray(v, np.float64) ** 2).sum()) for n, v in params.items() if n.endswith('.kernel'))


def loss_and_grads(spec, params, x, y, loss_cfg=None, training=True, n_replicas=1):
    """One replica's forward + loss + backward.  Keras semantics: scalar loss = mean over the batch of the
    per-sample losses (SUM_OVER_BATCH_SIZE) / n_replicas, plus the L2 regulariser / n_replicas."""
    loss_cfg = loss_cfg or {}
    logits, new_state, backward, _ = forward(spec, params, x, training=training)
    per, dper = weighted_crossentropy(y, logits, **loss_cfg)
    B = x.shape[0]
    loss = float(per.mean(dtype=np.float64)) / n_replicas + l2_penalty(spec, params) / n_replicas
    _, grads = backward(dper / logits.dtype.type(B * n_replicas))
    if spec.l2:
        for n in grads:
            if n.endswith('.kernel'):
                grads[n] = grads[n] + logits.dtype.type(2.0 * spec.l2 / n_replicas) * params[n]
    return loss, grads, logits, new_state


# --------------------------------------------------------------------------------------
# optimizer (engine.py:276-284) and LR schedule (deploy_options.yaml:3)
# --------------------------------------------------------------------------------------

def adam_step(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """Keras Adam (non-amsgrad) [TF-2.6 OptimizerV2]: t is the 1-based iteration;
    theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)."""
    out = OrderedDict()
    for n, g in grads.items():
        dt = params[n].dtype.type
        lr_t = dt(lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
        m[n] = m.get(n, 0) * dt(beta1) + g * dt(1 - beta1)
        v[n] = v.get(n, 0) * dt(beta2) + (g * g) * dt(1 - beta2)
        out[n] = params[n] - lr_t * m[n] / (np.sqrt(v[n]) + dt(eps))
    for n in params:
        if n not in out:
            out[n] = params[n]
    return out


def lr_schedule(step, base=0.001, decay=0.96, every=1000):
    """deploy_options.yaml:3: lambda epoch, current_lr: 0.001 * 0.96 ** (epoch // 1000)."""
    return base * decay ** (step // every)


def train_step(spec, params, m, v, t, x, y, lr, loss_cfg=None):
    """One full optimizer step on one replica; returns (loss, new params (incl. BN moving stats), grads, logits)."""
    loss, grads, logits, new_state = loss_and_grads(spec, params, x, y, loss_cfg, training=True)
    new_params = adam_step(params, grads, m, v, t, lr)
    for n, val in new_state.items():
        new_params[n] = val
    return loss, new_params, grads, logits


def predict(spec, params, x):
    """training=False forward: probabilities [B,H,W,1] = sigmoid(logits) (unet.py:241-244, 279-282)."""
    logits, _, _, _ = forward(spec, params, x, training=False)
    return 1.0 / (1.0 + np.exp(-logits)), logits


# --------------------------------------------------------------------------------------
# synthetic data
# --------------------------------------------------------------------------------------

def synthetic_batch(B, H, W, C, seed_x=0, seed_y=1, empty_first=False):
    """x = uint8/255 like data.py:205-206; y = union of 0-3 filled discs per slice."""
    rx = np.random.default_rng(seed_x)
    ry = np.random.default_rng(seed_y)
    x = (rx.integers(0, 256, size=(B, H, W, C), dtype=np.uint8) / np.float32(255.0)).astype(np.float32)
    y = np.zeros((B, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    s = min(H, W) / 512.0
    for b in range(B):
        n = int(ry.integers(0, 4))
        if b == 0 and not empty_first:
            n = max(n, 1)
        if b == 0 and empty_first:
            n = 0
        for _ in range(n):
            r = ry.uniform(8, 40) * s
            cy, cx = ry.uniform(64, 448, 2) * s
            y[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1.0
    return x, y
