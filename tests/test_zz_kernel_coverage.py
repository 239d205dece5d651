"""-m gpu, runs last: every launch of the three single-GPU BASELINE configurations at their full sizes has met the oracle.

The full-size tests compare tuned kernels with other kernels of this library (generic vs tuned, fused vs per-layer, bf16 vs fp32);
that is parity with the oracle only if every kernel those plans launch is ALSO one that some oracle-compared test ran: the golden
fixtures (tests/test_parity_gpu.py), the float64-oracle tests of tests/test_engine_gpu.py (3-channel level, block-fused backward,
non-square LeakyReLU / L2, label smoothing, dense configurations at real widths), the bf16-emulating oracle.  Those tests register
their models' launch names (helpers.record_oracle_plan); here the three BASELINE plans are built and every name must be in that set.
A launch name carries the kernel variant -- the pixel-group / fused kernels in the name itself (shape, ride-alongs), the dense
implicit-GEMM kernels as `name#variant` (channel tile, wave count, operand storage; DeviceModel.plan(variants=True)) -- so a
kernel that only a full-size shape selects shows up as a missing name."""

import pytest

import helpers as Hp

pytestmark = pytest.mark.gpu

REQUIRED = {'test_golden_tuned_kernels', 'test_vector_alu_kernels_of_the_3_channel_level_against_oracle',
            'test_block_fused_backward_against_oracle', 'test_bf16_kernels_against_bf16_emulating_oracle',
            'test_dense_configs_at_real_widths_against_oracle', 'test_fp32_eight_wave_conv_kernels_against_oracle'}
BASELINE = [
    ('configs/unet.yaml', 'unet', 1, 8, 'f32', dict(n_filters_first=3, n_downsample=3, bn=False)),
    ('configs/unet_big.yaml', 'unet', 1, 4, 'bf16', dict(n_filters_first=64, n_downsample=4, bn=True)),
    ('configs/mulmo_unet.yaml', 'mulmo', 3, 8, 'f32', dict(n_filters_first=16, n_downsample=4, bn=True)),
]
# bookkeeping launches without arithmetic of their own that only the dry plan shows (the live step folds them into neighbours)
BOOKKEEPING = {'g_step_init', 'g_finalize_scalars'}


def test_every_kernel_of_the_baseline_plans_has_met_the_oracle(gpu):
    missing_tests = REQUIRED - Hp.ORACLE_TESTS
    if missing_tests:
        pytest.skip('partial run: the oracle tests %s did not run in this session' % sorted(missing_tests))
    gaps = {}
    for name, arch, C, B, dtype, opts in BASELINE:
        m = gpu.DeviceModel(arch, C, 512, 512, B, rate=2, kernel_size=3, conv_stride=1, padding='same', dtype=dtype, **opts)
        plan = set(r[0] for r in m.plan(variants=True))
        m.close()
        assert len(plan) >= 10, plan
        print(name, len(plan), 'distinct launches:', ' '.join(sorted(plan)))
        gap = sorted(plan - Hp.ORACLE_KERNELS - BOOKKEEPING)
        if gap:
            gaps[name] = gap
    print('kernel coverage: %d launch names registered by %d oracle tests; BASELINE plans closed' % (len(Hp.ORACLE_KERNELS), len(Hp.ORACLE_TESTS)))
    assert not gaps, 'launches of the BASELINE plans that no oracle-compared test ran: %s' % gaps
