"""Independent torch-CPU (autograd) statement of the same network, used ONLY to pin the numpy oracle
(tests/test_oracle.py, tests/golden/make_golden.py).  Dev-time checker: not imported by the product.

Layout transposes: Conv2D HWIO -> OIHW; Conv2DTranspose [kh,kw,Cout,Cin] -> [Cin,Cout,kh,kw]."""

import numpy as np
import torch
import torch.nn.functional as F


def _t(a, dtype):
    return torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True)


def run(spec, params, x, y, loss_cfg, training, dtype=torch.float64):
    """returns dict(loss, logits NHWC, grads {name: ndarray}, state {name: ndarray})"""
    P = {n: _t(v, dtype) for n, v in params.items()}
    xt = torch.tensor(x, dtype=dtype).permute(0, 3, 1, 2)
    state = {}

    def act(t):
        return F.leaky_relu(t, spec.alpha) if spec.alpha else F.relu(t)

    def conv(prefix, t, activation=True):
        w = P[prefix + '.kernel'].permute(3, 2, 0, 1)
        pad = (spec.k - 1) // 2 if (spec.padding == 'same' and w.shape[-1] > 1) else 0
        t = F.conv2d(t, w, P[prefix + '.bias'], padding=pad)
        return act(t) if activation else t

    def bn(prefix, t):
        g, b = P[prefix + '.gamma'], P[prefix + '.beta']
        mm, mv = P[prefix + '.moving_mean'], P[prefix + '.moving_variance']
        if training:
            n = t.shape[0] * t.shape[2] * t.shape[3]
            mean = t.mean((0, 2, 3))
            var = t.var((0, 2, 3), unbiased=False)
            state[prefix + '.moving_mean'] = (mm * 0.99 + mean * 0.01).detach().numpy()
            state[prefix + '.moving_variance'] = (mv * 0.99 + var * (n / max(n - 1, 1)) * 0.01).detach().numpy()
        else:
            mean, var = mm, mv
        return (t - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-3) * g[None, :, None, None] \
            + b[None, :, None, None]

    skips_all, bottoms = [], []
    for e in range(spec.n_encoders()):
        enc = 'encoder%d' % e if spec.arch == 'mulmo' else 'encoder'
        t = xt[:, e:e + 1] if spec.arch == 'mulmo' else xt
        skips = []
        for i in range(spec.n_down):
            p = '%s.down%d' % (enc, i)
            for j in range(spec.n_conv):
                t = conv('%s.conv%d' % (p, j), t)
                if spec.bn:
                    t = bn('%s.bn%d' % (p, j), t)
            skips.append(t)
            t = F.max_pool2d(t, spec.rate, spec.rate)
            if spec.bn:
                t = bn(p + '.pool_bn', t)
        skips_all.append(skips)
        bottoms.append(t)
    t = torch.cat(bottoms, 1) if spec.arch == 'mulmo' else bottoms[0]
    ref = skips_all[spec.reference_index if spec.arch == 'mulmo' else 0]
    for u in range(spec.n_down):
        p = 'decoder.up%d' % u
        w = P[p + '.tconv.kernel'].permute(3, 2, 0, 1)
        t = F.conv_transpose2d(t, w, P[p + '.tconv.bias'], stride=spec.rate)
        if spec.bn:
            t = bn(p + '.tconv_bn', t)
        r = ref[spec.n_down - 1 - u]
        gh, gw = (r.shape[2] - t.shape[2]) // 2, (r.shape[3] - t.shape[3]) // 2
        t = torch.cat([t, r[:, :, gh:gh + t.shape[2], gw:gw + t.shape[3]]], 1)
        for j in range(spec.n_conv):
            t = conv('%s.conv%d' % (p, j), t)
            if spec.bn:
                t = bn('%s.bn%d' % (p, j), t)
    logits = F.conv2d(t, P['head.kernel'].permute(3, 2, 0, 1), P['head.bias'])

    yt = torch.tensor(y, dtype=dtype)
    cfg = dict(weight=None, weight_add=0.0, weight_mul=1.0)
    cfg.update(loss_cfg or {})
    w_ = cfg['weight']
    if w_ is None:
        pr = float(yt.sum() / yt.numel())
        w_ = 1.0 / pr if pr > 0 else 1.0
    w_ = cfg['weight_mul'] * w_ + cfg['weight_add']
    mask = yt * (w_ - 1.0) + 1.0
    bce = F.binary_cross_entropy_with_logits(logits[:, 0], yt, weight=mask, reduction='none')
    loss = bce.mean((1, 2)).mean()
    if spec.l2:
        loss = loss + spec.l2 * sum((v ** 2).sum() for n, v in P.items() if n.endswith('.kernel'))
    loss.backward()
    grads = {n: v.grad.numpy() for n, v in P.items() if v.grad is not None}
    return dict(loss=float(loss), logits=logits.detach().permute(0, 2, 3, 1).numpy(), grads=grads, state=state)
