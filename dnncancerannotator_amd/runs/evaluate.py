"""`annotator evaluate` -- annotator/runs/evaluate.py:21-91: evaluate every checkpoint under save_path/checkpoints."""

import os

from .. import engine, load
from .train import make_dataset


def evaluate(save_path, data_path, tag, config=None, avoid_overwrite=False, export_path=None, export_images=False,
             export_csv=False, visualize_sensitivity=False, min_interval=1, step_range=None, overlay=False,
             skip_visualization=False, export_casewise_metrics=False):
    saved_config = load.load_config(os.path.join(save_path, 'options.yaml'))['config']
    if config:
        config = load._apply_config(saved_config, load.load_config(config))
    else:
        config = saved_config
    ds = make_dataset(data_path, config.get('data_options', {}).get('eval', {}), training=False)
    model = engine.TFKerasModel(config)
    return model.eval(ds, viz_ds=None, tag=tag, save_path=os.path.join(save_path), avoid_overwrite=avoid_overwrite,
                      export_path=export_path, export_images=export_images, export_csv=export_csv,
                      visualize_sensitivity=visualize_sensitivity, min_interval=min_interval, step_range=step_range,
                      overlay=overlay, export_casewise_metrics=export_casewise_metrics)
