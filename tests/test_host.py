"""CPU: host-side mirror of the reference's config / CLI / registry / checkpoint-discovery surface."""

import json
import os

import numpy as np
import pytest
import yaml

from dnncancerannotator_amd import distributed, dump, engine, load, losses, metrics, models
from dnncancerannotator_amd.__main__ import build_parser

UNET_YAML = dict(model='UNetAnnotator', model_options=dict(n_filters_first=3, n_downsample=3, rate=2, kernel_size=3,
                                                           conv_stride=1, bn=False, padding='same'))
DEPLOY = {'deploy_options': {'optimizer': 'adam',
                             'LearningRateScheduler': 'lambda epoch, current_lr: 0.001 * 0.96 ** (epoch // 1000)',
                             'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                             'enable_multigpu': False}}


def _write(tmp_path, name, obj):
    p = tmp_path / name
    p.write_text(yaml.safe_dump(obj))
    return str(p)


def test_load_config_overlay_and_dotted_keys(tmp_path):
    base = _write(tmp_path, 'unet.yaml', UNET_YAML)
    dep = _write(tmp_path, 'deploy.yaml', DEPLOY)
    multi = _write(tmp_path, 'multigpu.yaml', {'deploy_options.enable_multigpu': True})        # multigpu.yaml:1
    leaky = _write(tmp_path, 'leaky.yaml', {'model_options.activation': {'class_name': 'LeakyReLU', 'config': {'alpha': 0.3}}})
    new = _write(tmp_path, 'new.yaml', {'data_options.train.batch_size': 28})                     # creates the path
    cfg = load.load_config([base, dep, multi, leaky, new])
    assert cfg['model'] == 'UNetAnnotator'
    assert cfg['deploy_options']['enable_multigpu'] is True and cfg['deploy_options']['optimizer'] == 'adam'
    assert cfg['model_options']['activation']['config']['alpha'] == 0.3
    assert cfg['data_options'] == {'train': {'batch_size': 28}}
    assert load.load_config(base) == UNET_YAML                       # a single str is accepted (load.py:32)
    j = tmp_path / 'c.json'
    j.write_text(json.dumps({'a': 1}))
    assert load.load_config(str(j)) == {'a': 1}
    with pytest.raises(NotImplementedError):
        load._load_config_single(str(tmp_path / 'c.toml'))
    with pytest.raises(AssertionError):
        load.load_config([])


def test_dump_options_never_overwrites(tmp_path):
    p = str(tmp_path / 'run' / 'options.yaml')
    first = dump.dump_options(p, config={'a': 1}, save_path='s', data_path=['d'])
    second = dump.dump_options(p, config={'a': 2}, save_path='s', data_path=['d'])
    assert first == p and os.path.basename(second) == 'options_.yaml'           # dump.py:30-33
    assert load.load_config(p)['config'] == {'a': 1}
    assert load.load_config(second)['config'] == {'a': 2}
    third = dump.dump_options(p, config={'a': 3})
    assert os.path.basename(third) == 'options__.yaml' and load.load_config(p)['config'] == {'a': 1}
    with pytest.raises(NotImplementedError):
        dump.dump_options(str(tmp_path / 'run' / 'options.toml'), a=1)
    jp = dump.dump_options(str(tmp_path / 'run' / 'o.json'), a=[1, 2])
    assert load.load_config(jp) == {'a': [1, 2]}


def test_dump_train_results_formats(tmp_path):
    import pickle
    import yaml

    class H:
        epoch, history, params, model = [0, 1], {'loss': [2.0, 1.5]}, {'epochs': 2}, object()
    p = dump.dump_train_results(str(tmp_path / 'r' / 'results.pkl'), H())
    assert pickle.load(open(p, 'rb')) == {'epoch': [0, 1], 'history': {'loss': [2.0, 1.5]}, 'params': {'epochs': 2}, 'model': 'object'}
    y = dump.dump_train_results(str(tmp_path / 'r' / 'results.yaml'), H(), format_='YAML')
    assert yaml.safe_load(open(y))['history'] == {'loss': [2.0, 1.5]}
    with pytest.raises(NotImplementedError):
        dump.dump_train_results(str(tmp_path / 'r' / 'x'), H(), format_='csv')


def test_loss_and_model_registries():
    l = losses.get({'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}})
    assert l.device_cfg() == dict(weight=None, weight_add=0.0, weight_mul=3.0, label_smoothing=False, label_smoothing_filter_size=6,
                                  label_smoothing_sigma=3.0)
    assert losses.get('weighted_crossentropy').weight_mul == 1.0
    with pytest.raises(ValueError):
        losses.get({'class_name': 'Nope'})
    assert losses.get({'class_name': 'WeightedCrossentropy', 'config': {'label_smoothing': True}}).device_cfg()['label_smoothing'] is True
    with pytest.raises(ValueError):
        losses.get({'class_name': 'WeightedCrossentropy', 'config': {'label_smoothing': True, 'label_smoothing_sigma': 0}})
    m = getattr(models, 'UNetAnnotator')(**UNET_YAML['model_options'])
    assert m.get_config()['n_filters_first'] == 3 and m.arch == 'unet'
    assert getattr(models, 'MulmoUNetAnnotator')(**UNET_YAML['model_options']).arch == 'mulmo'
    assert models._solve_activation({'class_name': 'LeakyReLU', 'config': {'alpha': 0.3}}) == 0.3
    assert models._solve_regularizer({'class_name': 'L2', 'config': {'l2': 0.01}}) == 0.01
    with pytest.raises(NotImplementedError):
        models.MultiResUnet()
    with pytest.raises(ValueError):
        models._solve_activation('tanh')


class _FakeDevice:
    """stands in for DeviceModel.pixel_confusion: counts on the CPU (prob > t, label > 0.5)"""

    def __init__(self, prob):
        self.prob = prob

    def pixel_confusion(self, y, thresholds):
        out = []
        for t in thresholds:
            pp, yy = self.prob > t, y > 0.5
            out.append((float((pp & yy).sum()), float((pp & ~yy).sum()), float((~pp & yy).sum()), float((~pp & ~yy).sum())))
        return out


def test_pixel_metrics_match_definitions():
    rng = np.random.default_rng(0)
    y = (rng.random((2, 16, 16)) < 0.3).astype(np.float32)
    prob = np.clip(0.6 * y + 0.5 * rng.random(y.shape), 0, 1).astype(np.float32)
    dev = _FakeDevice(prob)
    spec = [{'Precision': {'thresholds': 0.8, 'name': 'pixel/precision'}}, {'Recall': {'thresholds': 0.8, 'name': 'pixel/recall'}},
            {'FBetaScore': {'thresholds': 0.8, 'beta': 1.0, 'name': 'pixel/F1-score'}},
            {'AUC': {'curve': 'ROC', 'name': 'pixel/AUROC', 'num_thresholds': 150}},
            {'AUC': {'curve': 'PR', 'name': 'pixel/AUPRC', 'num_thresholds': 150}},
            {'RegionBasedRecall': {'thresholds': 0.8, 'IoU_threshold': 0.3}}]
    ms = [metrics.solve_metric(s) for s in spec]
    assert ms[-1] is None                                    # region metrics: outside the hot path
    for m in ms[:-1]:
        m.update_state(dev, y)
    pp, yy = prob > 0.8, y > 0.5
    tp, fp, fn = (pp & yy).sum(), (pp & ~yy).sum(), (~pp & yy).sum()
    P, R = tp / (tp + fp), tp / (tp + fn)
    assert abs(ms[0].result() - P) < 1e-12 and abs(ms[1].result() - R) < 1e-12
    assert abs(ms[2].result() - 2 * P * R / (P + R + 1e-7)) < 1e-12          # metrics.py:60
    from sklearn.metrics import roc_auc_score, average_precision_score
    assert abs(ms[3].result() - roc_auc_score(yy.ravel(), prob.ravel())) < 2e-2
    assert abs(ms[4].result() - average_precision_score(yy.ravel(), prob.ravel())) < 3e-2
    with pytest.raises(ValueError):
        metrics.solve_metric({'Nope': {}})


def test_cli_flag_surface():
    p = build_parser()
    a = p.parse_args(['train', '--config', 'a.yaml', 'b.yaml', '--save_path', 's', '--data_path', 'd1', 'd2', '--max_steps', '7',
                      '--early_stop_steps', '3', '--validate', '--val_data_path', 'v'])
    assert a.config == ['a.yaml', 'b.yaml'] and a.data_path == ['d1', 'd2'] and a.max_steps == 7 and a.save_freq == 500
    assert a.validate and a.val_data_path == ['v'] and not a.visualize and not a.profile
    e = p.parse_args(['evaluate', '--save_path', 's', '--data_path', 'd', '--tag', 't', '--step_range', '10', '20', '--export_csv'])
    assert e.step_range == [10, 20] and e.min_interval == 1 and e.export_csv and e.config is None
    with pytest.raises(SystemExit):
        p.parse_args(['train', '--config', 'a.yaml'])


def test_engine_config_checks_and_ckpt_discovery(tmp_path):
    cfg = dict(UNET_YAML, **DEPLOY)
    m = engine.TFKerasModel(cfg)
    assert m.get_config() == cfg and m.enable_multigpu is False
    assert m.loss.weight_mul == 3.0 and m.learning_rate_scheduler.startswith('lambda epoch')
    for missing in ('model', 'model_options', 'deploy_options'):
        bad = {k: v for k, v in cfg.items() if k != missing}
        with pytest.raises(AssertionError):
            engine.TFKerasModel(bad)
    for name in ('ckpt-5.index', 'ckpt-5.data-00000-of-00001', 'ckpt-100.index', 'ckpt-20.index', 'checkpoint', 'ckpt-x.index'):
        (tmp_path / name).write_text('')
    ck = m.get_ckpts(str(tmp_path))
    assert list(ck.keys()) == [5, 20, 100] and ck[20] == str(tmp_path / 'ckpt-20')        # engine.py:55-65
    no_multi = dict(cfg, deploy_options={k: v for k, v in cfg['deploy_options'].items() if k != 'enable_multigpu'})
    assert engine.TFKerasModel(no_multi).enable_multigpu is True                          # default on (engine.py:260)


def test_distributed_context_and_shards():
    c = distributed.context({'RANK': '3', 'LOCAL_RANK': '1', 'WORLD_SIZE': '8', 'MASTER_PORT': '29500'})
    assert (c.rank, c.local_rank, c.world) == (3, 1, 8)
    assert distributed.context({}).world == 1
    assert distributed.shard_bounds(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        distributed.shard_bounds(10, 0, 4)
    with pytest.raises(ValueError):
        distributed.context({'RANK': '2', 'WORLD_SIZE': '2'})
