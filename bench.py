#!/usr/bin/env python3
"""bench.py -- MRI slices/s of one train step (forward + weighted-BCE + backward + Adam) of configs/unet.yaml on
synthetic 512x512x1 batches, 8 slices per GPU, fp32, on N MI355X (weak scaling: global batch = 8 N).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; gradients are summed by ONE RCCL all-reduce per step inside libdnnca (no torch in the workers:
torch bundles its own ROCm runtime, which must not share a process with libdnnca).  The RCCL unique id travels through
a file keyed by the launcher's pid.  Rank 0 prints one JSON line.

The line's `value` is the headline configuration (BASELINE.json configs[1]).  At N = 1 the same run also times the other two
single-GPU configurations of BASELINE.json (configs[2] unet_big bf16 B=4, configs[3] mulmo_unet f32 B=8) for a few steps each
and reports them under `other_workloads` (informative, never `value`).  With fewer than 100 steps the timed region is repeated
(`repeats`, `region_ms`) and the line carries the median region: `steps` / `ms_per_step` describe ONE region of exactly K steps.

DNNCA_FORCE_COMM=1 rehearses the N > 1 sequence on one GPU: id exchange through the file, a one-rank RCCL communicator,
barriers and the max-reduction of the wall time through it (tests/test_engine_gpu.py::test_rccl_one_rank_rehearsal).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dnncancerannotator_amd import device as dev                      # noqa: E402
from dnncancerannotator_amd import distributed                        # noqa: E402
from dnncancerannotator_amd.synthetic import synthetic_batch         # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md:36 (spec; 6.29 TB/s measured float4 copy)
H = W = 512
UNET_YAML = dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same')
# the single-GPU configurations of BASELINE.json: configs[1] is the metric's; [2] and [3] ride along under `other_workloads`
WORKLOADS = {
    'unet': dict(arch='unet', C=1, batch=8, dtype='f32', opts=UNET_YAML, name='configs/unet.yaml'),
    'unet_big': dict(arch='unet', C=1, batch=4, dtype='bf16', name='configs/unet_big.yaml',
                     opts=dict(n_filters_first=64, n_downsample=4, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')),
    'mulmo_unet': dict(arch='mulmo', C=3, batch=8, dtype='f32', name='configs/mulmo_unet.yaml',
                       opts=dict(n_filters_first=16, n_downsample=4, rate=2, kernel_size=3, conv_stride=1, bn=True, padding='same')),
}
PROF_STEPS = 5
SETTLE_MS = 100.0              # un-timed steps in front of the timed regions, in addition to --warmup: at least 20 steps and at least
SETTLE_MIN_STEPS = 20          # this much device time -- the shader / memory clocks are still rising after the first ~30 steps
CPU_THREADS = 16               # the GPU box gives one GPU's job 16 host cores; numpy's BLAS pool is pinned to that many


def cpu_baseline(sample_steps=5):
    """The numpy oracle (a port: TensorFlow, the reference's CPU back-end, is not installed anywhere) timed on this host on a
    bounded sample of the same workload: `sample_steps` train steps of one 8-slice batch (C2) and of one 1-slice batch (C1,
    BASELINE.md's CPU-baseline plan).  The BLAS / OpenMP pools are limited to CPU_THREADS threads and that number is what
    `cores` reports; the rest of numpy (elementwise passes, im2col copies) is single-threaded."""
    from oracle import unet_oracle as O
    spec = O.ModelSpec('unet', 1, **UNET_YAML)
    x, y = synthetic_batch(8, H, W, 1)
    cfg = dict(weight_mul=3.0)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=CPU_THREADS)
        threads = CPU_THREADS
    except Exception:
        limiter, threads = None, 1

    def timed(batch, steps):
        params = O.init_params(spec, seed=2)
        m, v = {}, {}
        O.train_step(spec, params, m, v, 1, x[:1], y[:1], 1e-3, cfg)      # page in / warm BLAS
        t0 = time.time()
        for t in range(steps):
            _, params, _, _ = O.train_step(spec, params, m, v, t + 1, x[:batch], y[:batch], 1e-3, cfg)
        return steps * batch / (time.time() - t0)

    c2 = timed(8, sample_steps)
    c1 = timed(1, 2 * sample_steps)
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, 'restore_original_limits') else None
    return {'value': round(c2, 3), 'unit': 'slices/s', 'cores': int(threads), 'kind': 'port',
            'sample': '%d train steps of one 8x%dx%dx1 batch (C2) -- numpy oracle, fp32, BLAS pool limited to %d threads on a host '
                      'with %d cores; TensorFlow (the reference CPU path) is not installed' % (sample_steps, H, W, threads, os.cpu_count()),
            'c1_batch1': {'value': round(c1, 3), 'unit': 'slices/s', 'sample': '%d train steps of one 1x%dx%dx1 batch (C1)' % (2 * sample_steps, H, W)}}


def time_workload(wname, steps, warmup, ctx, comm, generic=False, repeats=1, host_leg=False):
    """Builds the workload's model on this rank's GPU and times `repeats` regions of exactly `steps` train steps on a batch that is
    resident in HBM, each region bracketed by a barrier + device synchronisation on both sides; the wall time of a region is the
    maximum over the ranks.  Returns the fields of the JSON line that describe this workload (rank 0's view)."""
    rank, world = ctx.rank, ctx.world
    wl = WORKLOADS[wname]
    C, B = wl['C'], wl['batch']
    model = dev.DeviceModel(wl['arch'], C, H, W, B, force_generic=generic, dtype=wl['dtype'], **wl['opts'])
    model.init_glorot(seed=2)          # same weights on every rank (random-init weights of the named architecture)
    if comm:                           # (one GPU, not forced: no communicator -- the single-replica step keeps its fused fold + Adam launch)
        uid = distributed.exchange_unique_id(ctx, dev.DeviceModel, force=True, tag=wname)     # 128-byte RCCL id through a file keyed by the launcher's pid
        model.comm_init(rank, world, uid)

    # rank-local shard of the global batch, resident in HBM before the timed region
    x, y = synthetic_batch(B, H, W, C, seed_x=100 + rank, seed_y=200 + rank)
    xb, yb = dev.DeviceBuffer(x), dev.DeviceBuffer(y)
    cfg = model.loss_cfg(weight_mul=3.0)            # configs/additionals/deploy_options.yaml:5-7
    lr = 1e-3

    def barrier():
        model.sync()
        if comm:
            model.comm_allreduce([0.0])
        model.sync()

    tw = time.perf_counter()
    for _ in range(warmup):
        model.train_step_dev(xb, yb, B, lr, cfg)
    model.sync()
    tw = 1e3 * (time.perf_counter() - tw) / max(warmup, 1)          # ms per warm-up step (an over-estimate: first launches)

    # which kernel dominates?  a few fully instrumented steps outside the timed region
    model.profile_reset()
    model.profile_enable(1)
    for _ in range(PROF_STEPS):
        model.train_step_dev(xb, yb, B, lr, cfg)
    model.sync()
    full_table = list(model.profile())
    # (only kernels that move data or compute: under a profiler the bracket of a bookkeeping launch can absorb one-off costs)
    table = sorted((r for r in model.profile() if r[3] > 0 or r[4] > 0), key=lambda r: -r[2])
    dominant = table[0][0]
    dominant_per_step = table[0][1] / float(PROF_STEPS)          # launches of the dominant kernel per step
    model.profile_enable(0)
    model.profile_reset()
    # HIP events around the dominant kernel only, on the launch stream, in every 4th step of the timed region (a bracket
    # costs ~3 us of dispatch: bracketing every launch would take 1.5 % off `value`)
    model.profile_enable(2, focus=dominant, period=4)
    # settle: the regions' times kept falling by ~1 % per region after 5 + 5 steps (clocks); `steps` / `warmup` stay as given
    settle_steps = max(SETTLE_MIN_STEPS, int(SETTLE_MS / max(tw, 1e-3)))
    for _ in range(settle_steps):
        model.train_step_dev(xb, yb, B, lr, cfg)
    model.sync()
    model.profile_reset()

    regions, ev_regions = [], []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        model.timer_start()
        for _ in range(steps):
            model.train_step_dev(xb, yb, B, lr, cfg)
        ev_ms = model.timer_stop()
        barrier()
        wall = time.perf_counter() - t0
        regions.append(float(model.comm_allreduce([wall], op='max')[0]) if comm else wall)
        ev_regions.append(ev_ms)
    out = model.last_step_out()
    prof = {r[0]: r for r in model.profile()}
    model.profile_enable(0)

    res = None
    if rank == 0:
        order = sorted(range(repeats), key=lambda i: regions[i])
        mid = order[repeats // 2]                    # the median region (repeats is odd or 1)
        elapsed = regions[mid]
        ms_per_step = 1e3 * elapsed / steps
        value = world * B * steps / elapsed
        name, launches, total_ms, bytes_per, flops_per = prof.get(dominant, table[0])      # (table[0]: the instrumented steps)
        avg_ms_region = total_ms / max(launches, 1)           # HIP-event brackets inside the timed region
        avg_ms_alone = table[0][2] / max(table[0][1], 1)      # the PROF_STEPS serialised, fully bracketed steps in front of it
        # Which duration prices the kernel?  configs/unet.yaml runs on one stream: the timed region's brackets (the contract's
        # "measured live over the timed region").  The dense configurations run their weight gradients on a side stream beside the
        # main chain, so a bracket in the timed region measures the kernel while it SHARES the chip (`avg_launch_us_overlapped`);
        # the kernel's own roofline fraction comes from the serialised steps (`avg_launch_us_alone`), the same figure
        # `roofline_all` / `top_kernels` carry.
        avg_ms = avg_ms_region if wname == 'unet' else avg_ms_alone
        mfma_peak = 2500.0 if wl['dtype'] == 'bf16' else 157.3        # TFLOP/s dense, MI355X_MICROARCH.md:42-43

        def kernel_peak(kname):
            """MFMA peak a kernel is priced against, in the units its flops are counted in (fp32 multiply-adds x 2).  The ig3x_*
            kernels (csrc/kernels_ig3x.hip) compute fp32 products as six bf16 products on the bf16 matrix pipe: 2.5 PFLOP/s / 6."""
            return 2500.0 / 6.0 if kname.startswith('ig3x_') else mfma_peak
        if wname == 'unet':                  # HBM-bound (AI ~ 9 FLOP/B): algorithmic bytes of the launch / its duration
            bound, unit, peak = 'hbm', 'GB/s', HBM_PEAK_GBS
            achieved = bytes_per / (avg_ms * 1e-3) / 1e9
        else:                                 # dense contractions: algorithmic FLOPs / duration against the MFMA peak
            bound, unit, peak = 'mfma', 'TFLOP/s', mfma_peak
            achieved = flops_per / (avg_ms * 1e-3) / 1e12
        # every kernel of the step (north_star: "achieved fraction of HBM/MFMA roofline reported per kernel"), from the
        # PROF_STEPS fully bracketed steps before the timed region (a HIP-event bracket adds ~3 us of dispatch to a launch,
        # so the short kernels read low here; the rocprofv3 summary under profiles/ has the un-bracketed durations).
        # Each kernel is priced against the roofline that bounds it: HBM when its arithmetic intensity is below the ridge.
        roofline_all, step_bytes, step_flops = [], 0.0, 0.0
        for kname, klaunch, kms, kbytes, kflops in sorted(full_table, key=lambda r: -r[2]):
            if klaunch == 0:
                continue
            us = 1e3 * kms / klaunch
            step_bytes += kbytes * klaunch / PROF_STEPS
            step_flops += kflops * klaunch / PROF_STEPS
            kpeak = kernel_peak(kname)
            ridge = kpeak * 1e12 / (HBM_PEAK_GBS * 1e9)
            hbm = kbytes <= 0 or kflops / max(kbytes, 1.0) < ridge
            ach = (kbytes / (us * 1e-6) / 1e9) if hbm else (kflops / (us * 1e-6) / 1e12)
            roofline_all.append({'kernel': kname, 'launches_per_step': round(klaunch / PROF_STEPS, 2), 'avg_us': round(us, 2),
                                 'algorithmic_bytes': kbytes, 'flops': kflops, 'bound': 'hbm' if hbm else 'mfma',
                                 'achieved': round(ach, 1), 'unit': 'GB/s' if hbm else 'TFLOP/s',
                                 'peak': HBM_PEAK_GBS if hbm else round(kpeak, 1),
                                 'frac': round(ach / (HBM_PEAK_GBS if hbm else kpeak), 4)})
        step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
        frac = achieved / peak
        if wname != 'unet':                  # ONE figure per kernel: the dense legs quote the dominant kernel's `roofline_all` entry
            e = next(r for r in roofline_all if r['kernel'] == dominant)
            bound, unit, achieved, frac, peak = e['bound'], e['unit'], e['achieved'], e['frac'], e['peak']
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
        if os.path.exists(tpath) and wname == 'unet':
            with open(tpath) as f:
                traffic = json.load(f).get(dominant)
            if traffic is not None:
                traffic_source = ('profiles/roofline_traffic.json -- FETCH_SIZE + WRITE_SIZE per launch from the builder\'s rocprofv3 '
                                  '--pmc passes over this same command (tools/collect_profiles.sh), not measured in this run')
        res = {
            'value': round(value, 2), 'ms_per_step': round(ms_per_step, 4), 'dtype': wl['dtype'], 'batch': B, 'channels': C,
            'workload': '%s train step (fwd + weighted BCE + bwd + Adam), batch %d x 512x512x%d per GPU, %s, random-init weights'
                        % (wl['name'], B, C, wl['dtype']),
            'repeats': repeats, 'region_ms': [round(1e3 * r, 3) for r in regions], 'settle_steps': settle_steps,
            'roofline': {'bound': bound, 'kernel': dominant, 'achieved': round(achieved, 1), 'peak': peak,
                         'unit': unit, 'frac': round(frac, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                         'launches': int(launches), 'avg_launch_us': round(avg_ms * 1e3, 2),
                         'avg_launch_us_alone': round(avg_ms_alone * 1e3, 2), 'avg_launch_us_overlapped': round(avg_ms_region * 1e3, 2),
                         'frac_from': 'timed region brackets' if wname == 'unet' else 'serialised steps (avg_launch_us_alone)',
                         'peak_note': ('fp32 results from six bf16 products per multiply-add on the bf16 matrix pipe (csrc/kernels_ig3x.hip): '
                                       'peak = 2.5 PFLOP/s / 6 in fp32-equivalent FLOP/s' if dominant.startswith('ig3x_') else None),
                         'algorithmic_bytes_per_launch': bytes_per, 'flops_per_launch': flops_per,
                         'share_of_step': round(avg_ms * dominant_per_step / (ms_per_step if ms_per_step > 0 else 1e9), 4)},
            # the whole step against the HBM roofline: sum of the launches' algorithmic bytes (every tensor read
            # once and written once per layer, backward = 2 x forward) / the measured step time
            'roofline_step': {'bound': 'hbm', 'algorithmic_bytes': step_bytes, 'flops': step_flops, 'achieved': round(step_gbs, 1),
                              'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(step_gbs / HBM_PEAK_GBS, 4),
                              'launches_per_step': round(sum(r[1] for r in full_table) / PROF_STEPS, 1)},
            'roofline_all': roofline_all,
            'hip_event_ms_per_step': round(ev_regions[mid] / steps, 4),
            'final_loss': round(float(out.loss), 6),
        }
        if host_leg:
            # informative, never `value`: the same steps with every batch starting in (pageable) host memory -- uploads into the
            # model's staging ring on the copy stream, step scalars read one step late (NOTES.md 4b; tools/e2e_rate.py)
            ring = model.staging()
            hb = [synthetic_batch(B, H, W, C, seed_x=300 + i, seed_y=400 + i) for i in range(2)]
            n_host, prev = max(20, steps // 2), None
            model.sync()
            th = time.perf_counter()
            for i in range(n_host):
                slot = i % ring.slots
                px, py = ring.upload(slot, hb[i % 2][0], hb[i % 2][1], wait=False)
                ring.train_step(slot, px, py, B, lr, cfg)
                if prev is not None:
                    ring.out(prev)
                prev = slot
            ring.out(prev)
            th = time.perf_counter() - th
            res['from_host_memory'] = {'value': round(B * n_host / th, 1), 'unit': 'slices/s', 'steps': n_host,
                                       'ms_per_step': round(1e3 * th / n_host, 4),
                                       'note': 'PCIe-inclusive: %.1f MB per step from pageable host arrays through the staging ring'
                                               % ((hb[0][0].nbytes + hb[0][1].nbytes) / 1e6)}
    model.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true', help='skip the informative unet_big / mulmo_unet legs')
    ap.add_argument('--generic', action='store_true', help='force the untuned generic kernels')
    ap.add_argument('--workload', default='unet', choices=sorted(WORKLOADS), help='default: the metric\'s configuration (unet)')
    args = ap.parse_args()

    ctx = distributed.context()          # RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run (one process per GPU)
    rank, local_rank, world = ctx.rank, ctx.local_rank, ctx.world
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE %d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    dev.init_device(local_rank)
    # N > 1: the communicator; DNNCA_FORCE_COMM=1: the same sequence on one rank (rehearsal: a sum over one rank is the identity)
    force_comm = world == 1 and os.environ.get('DNNCA_FORCE_COMM') == '1'
    if force_comm:
        os.environ['DNNCA_FORCE_RCCL'] = '1'
    comm = world > 1 or force_comm
    repeats = 1 if args.steps >= 100 else 3
    wl = WORKLOADS[args.workload]
    head = time_workload(args.workload, args.steps, args.warmup, ctx, comm, generic=args.generic, repeats=repeats,
                         host_leg=(world == 1 and args.workload == 'unet' and not comm))
    if rank == 0:
        line = {
            'metric': 'MRI slices/sec (fwd+bwd) %s 512x512 bs=%d' % (args.workload, wl['batch']), 'value': head['value'], 'unit': 'slices/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': head['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': wl['dtype'], 'data': 'synthetic',
            'config': {'workload': head['workload'], 'global_batch': world * wl['batch'], 'parallelism': 'dp%d' % world,
                       'kernels': 'generic' if args.generic else 'tuned', 'communicator': bool(comm)},
        }
        for k in ('repeats', 'region_ms', 'settle_steps', 'roofline', 'roofline_step', 'roofline_all', 'hip_event_ms_per_step', 'final_loss', 'from_host_memory'):
            if k in head:
                line[k] = head[k]
    if world == 1 and not comm and args.workload == 'unet' and not args.generic and not args.no_other_workloads:
        # BASELINE.json configs[2] and [3], a few steps each (informative; the model of the headline leg is closed by now)
        others = {}
        n_other = max(5, args.steps // 4)
        for wname in ('unet_big', 'mulmo_unet'):
            r = time_workload(wname, n_other, max(3, args.warmup // 2), ctx, False)
            owl = WORKLOADS[wname]
            others[wname] = {'metric': 'MRI slices/sec (fwd+bwd) %s 512x512 bs=%d' % (wname, owl['batch']), 'value': r['value'], 'unit': 'slices/s',
                             'ms_per_step': r['ms_per_step'], 'steps': n_other, 'dtype': r['dtype'], 'workload': r['workload'],
                             'arithmetic': ('fp32 tensors and results; the dense 3x3 convs (forward, data and weight gradient) form each fp32 product from '
                                            'six bf16 products of three-plane operands on the bf16 matrix pipe, fp32 accumulation (csrc/kernels_ig3x.hip; '
                                            'DNNCA_NO_X3=1 runs them on the fp32 matrix pipe)') if owl['dtype'] == 'f32' else
                                           'bf16 MFMA operands (RNE), fp32 master weights and accumulation, bf16-stored BatchNorm inputs / gradients',
                             'settle_steps': r['settle_steps'], 'region_ms': r['region_ms'],
                             'roofline': {k: r['roofline'][k] for k in ('bound', 'kernel', 'achieved', 'peak', 'peak_note', 'unit', 'frac', 'frac_from', 'avg_launch_us',
                                                                        'avg_launch_us_alone', 'avg_launch_us_overlapped', 'share_of_step')},
                             'roofline_step': r['roofline_step'],
                             'top_kernels': r['roofline_all'][:6]}
        line['other_workloads'] = others
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.workload == 'unet':
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    distributed.cleanup(ctx)


if __name__ == '__main__':
    main()
