"""CLI: `python3 -m dnncancerannotator_amd {train,evaluate} ...` (also reachable as `python3 -m annotator ...`).

Flag surface of the reference (README.md:16-120; runs/train.py:21-54, runs/evaluate.py:21-64), re-stated with argparse
because dsargparse (requirements.txt:11) is not available."""

import argparse
import logging
import sys


def build_parser(prog='python3 -m annotator'):
    parser = argparse.ArgumentParser(prog=prog, description='DNNAnnotator: DNN model to predict cancer segmentation (MI355X engine)')
    sub = parser.add_subparsers(dest='command', help='command')
    t = sub.add_parser('train', help='Train a model with specified configs.')
    t.add_argument('--config', nargs='+', required=True, help='configuration file path(s); later files overlay the first')
    t.add_argument('--save_path', required=True, help='where to save weights/configs/results')
    t.add_argument('--data_path', nargs='+', required=True, help='path to the data root dir')
    t.add_argument('--max_steps', type=int, required=True, help='max training steps')
    t.add_argument('--early_stop_steps', type=int, default=None, help='steps to train without improvements')
    t.add_argument('--save_freq', type=int, default=500, help='interval of checkpoints (default: 500 steps)')
    t.add_argument('--validate', action='store_true', help='also validate the model on the validation dataset')
    t.add_argument('--val_data_path', nargs='+', default=None, help='path to the validation dataset')
    t.add_argument('--visualize', action='store_true', help='should visualize results')
    t.add_argument('--profile', action='store_true', help='enable profiling')
    e = sub.add_parser('evaluate', help='Evaluate a model with specified configs for every checkpoints available.')
    e.add_argument('--save_path', required=True)
    e.add_argument('--data_path', nargs='+', required=True)
    e.add_argument('--tag', required=True, help='save tag')
    e.add_argument('--config', nargs='+', default=None)
    e.add_argument('--avoid_overwrite', action='store_true')
    e.add_argument('--export_path', default=None)
    e.add_argument('--export_images', action='store_true')
    e.add_argument('--export_csv', action='store_true')
    e.add_argument('--visualize_sensitivity', action='store_true')
    e.add_argument('--min_interval', type=int, default=1)
    e.add_argument('--step_range', type=int, nargs=2, default=None, help='"--step_range start end"')
    e.add_argument('--overlay', action='store_true')
    e.add_argument('--skip_visualization', action='store_true')
    e.add_argument('--export_casewise_metrics', action='store_true')
    return parser


def main(argv=None, prog='python3 -m annotator'):
    logging.basicConfig(level=logging.INFO, format='%(levelname)s %(message)s')
    parser = build_parser(prog)
    args = vars(parser.parse_args(argv))
    command = args.pop('command')
    if command == 'train':
        from .runs.train import train
        train(**args)
    elif command == 'evaluate':
        from .runs.evaluate import evaluate
        rows = evaluate(**args)
        for step, r in (rows or {}).items():
            print(step, dict(r))
    else:
        parser.print_help()
        return 2
    return 0


if __name__ == '__main__':
    sys.exit(main(prog='python3 -m dnncancerannotator_amd'))
