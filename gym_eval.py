#!/usr/bin/env python
"""gym_eval.py — drop-in for the reference's gym_eval.py (same flags, gym_eval.py:15-37): loads a full / tracker /
target checkpoint (reference names and keys), evaluates `--num-episodes` episodes with the argmax policy and reports
R_mean / R_std / EL_mean / EL_std / S_rate (success = episode length >= 500), optionally appending a CSV row
(gym_eval.py:122-141). The episodes run as one batch on the GPU."""
from __future__ import division
import os
os.environ["OMP_NUM_THREADS"] = "1"
import argparse
import logging

import numpy as np
import torch

from active_tracking_rl_amd import build
from active_tracking_rl_amd.environment import _spaces
from active_tracking_rl_amd.model import build_model
from active_tracking_rl_amd.test import evaluate
from active_tracking_rl_amd import registry
from active_tracking_rl_amd.utils import check_path, setup_logger

parser = argparse.ArgumentParser(description='A3C_EVAL')
parser.add_argument('--env', default='Track2D-BlockPartialNav-v0', metavar='ENV', help='environment to evaluate on')
parser.add_argument('--num-episodes', type=int, default=100, metavar='NE', help='how many episodes in evaluation')
parser.add_argument('--load-model-dir', default=None, metavar='LMD', help='full checkpoint')
parser.add_argument('--load-tracker', default=None, metavar='LCD', help='tracker checkpoint')
parser.add_argument('--load-target', default=None, metavar='LCD', help='target checkpoint')
parser.add_argument('--log-dir', default='logs/', metavar='LG', help='folder to save logs')
parser.add_argument('--csv', default=None, metavar='SV', help='write to csv')
parser.add_argument('--render', dest='render', action='store_true', help='(not supported on the batched path)')
parser.add_argument('--network', default='tat-maze-lstm', metavar='M', help='Model type to use')
parser.add_argument('--stack-frames', type=int, default=1, metavar='SF', help='Choose whether to stack observations')
parser.add_argument('--seed', type=int, default=1, metavar='S', help='random seed (default: 1)')
parser.add_argument('--gpu-id', type=int, default=0, help='GPU to use')
parser.add_argument('--obs', default='img', metavar='UE', help='unreal env')
parser.add_argument('--single', dest='single', action='store_true', help='single agent')
parser.add_argument('--rescale', dest='rescale', action='store_true', help='rescale image to [-1, 1]')
parser.add_argument('--rnn-out', type=int, default=128, metavar='LO', help='lstm output size')
parser.add_argument('--aux', default='reward', help='auxiliary task: reward/none')

if __name__ == '__main__':
    args = parser.parse_args()
    build.build()
    check_path(args.log_dir)
    name = '{}_mon_log'.format(args.env)
    setup_logger(name, os.path.join(args.log_dir, name))
    log = logging.getLogger(name)
    device = torch.device('cuda', max(args.gpu_id, 0))
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)
    for k, v in vars(args).items():
        log.info('{0}: {1}'.format(k, v))
    # the model is sized by the env's observation space, as in the reference (gym_eval.py:62-69): 13x13 crops for the
    # 'Partial' ids, the whole S x S map (S = 81 Maze, 82 Block/Empty) for the 'Full' ids
    sp = registry.spec(args.env)
    full_side = 81 if sp["map_type"] == "Maze" else 82
    obs_space, act_space = _spaces((full_side, full_side) if sp["obs_type"] == "Full" else (13, 13))
    model = build_model(obs_space, act_space, args, device).to(device)
    load = lambda path: torch.load(path, map_location=lambda storage, loc: storage)
    if args.load_model_dir is not None:
        model.load_state_dict(load(args.load_model_dir), strict=False)     # gym_eval.py:74-78
    if args.load_tracker is not None:
        model.player0.load_state_dict(load(args.load_tracker))            # :81-85
    if args.load_target is not None:
        model.player1.load_state_dict(load(args.load_target))             # :88-92
    args.gpu_ids = [device.index]
    rsum, length = evaluate(model, args.env, args, device, args.num_episodes)
    reward_mean, reward_std = rsum.mean(0), rsum.std(0)
    len_mean, len_std = length.mean(), length.std()
    success_rate = float((length >= 500).mean())
    log.info("El, {0}, R, {1}, R_mean: {2}, R_std: {3}, EL_mean: {4:.2f}, EL_std {5:.2f}, R_step: {6}, S_rate: {7}".format(
        int(length[-1]), rsum[-1], reward_mean, reward_std, len_mean, len_std, reward_mean / len_mean, success_rate))
    if args.csv is not None:
        import csv
        header = ['Env', 'Seed', 'R_mean', 'R_std', 'EL_mean', 'EL_std', 'S_rate']
        row = {'Env': args.env, 'Seed': args.seed, 'R_mean': float(reward_mean[0]), 'R_std': float(reward_std[0]),
               'EL_mean': float(len_mean), 'EL_std': float(len_std), 'S_rate': success_rate}
        new = not os.path.exists(args.csv)
        with open(args.csv, 'a') as f:
            w = csv.DictWriter(f, header)
            if new:
                w.writeheader()
            w.writerows([row])
