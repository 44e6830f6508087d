"""`python3 train.py cfg/X.cfg` -- same command line, cfg format and printed output as the reference's
entry point (reference train.py:15-67): one experiment per cfg section, header printed once, then
`section, mean, std`.  Environments come from this package's registry instead of gym/gym_flock, and the
learner runs on the MI355X kernels.
"""
from os import path
import configparser
import os
import random
import sys

import numpy as np
import torch

from multiagent_gnn_policies_amd import envs, parallel
from multiagent_gnn_policies_amd.learner.gnn_dagger import train_dagger
from multiagent_gnn_policies_amd.learner.gnn_cloning import train_cloning
from multiagent_gnn_policies_amd.learner.gnn_baseline import train_baseline


def run_experiment(args):
    env_name = args.get('env')
    env = envs.make(env_name, device='cuda:%d' % parallel.local_device_index())
    if isinstance(env.env, envs.FlockingRelativeEnv):
        env.env.params_from_cfg(args)

    # one seed for the four RNG streams, as in reference train.py:24-28 (+ rank under torchrun, so every
    # rank rolls out different episodes; weights are broadcast from rank 0 when the learner is built)
    rank, world, local_rank = parallel.init_from_env()
    seed = args.getint('seed') + rank
    env.seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)

    if not torch.cuda.is_available():
        raise RuntimeError("train.py needs an MI355X (HIP device); this framework has no CPU compute path")
    device = torch.device("cuda", parallel.local_device_index(local_rank))
    torch.cuda.set_device(device)

    alg = args.get('alg').lower()
    if alg == 'dagger':
        stats = train_dagger(env, args, device)
    elif alg == 'dagger_vec':
        # extension: device-resident vectorised DAGGER, `n_envs` parallel episodes per GPU (learner/vec_dagger.py)
        from multiagent_gnn_policies_amd.learner.vec_dagger import train_dagger_vec
        env.close()
        stats = train_dagger_vec(args, device, n_envs=args.getint('n_envs', fallback=64))
    elif alg == 'cloning':
        stats = train_cloning(env, args, device)
    elif alg == 'baseline':
        stats = train_baseline(env, args)
    else:
        raise Exception('Invalid algorithm/mode name')
    return stats


def main():
    fname = sys.argv[1]
    config_file = fname if path.isabs(fname) else path.join(path.dirname(path.abspath(__file__)), fname)
    if not path.exists(config_file):
        config_file = fname
    config = configparser.ConfigParser()
    config.read(config_file)

    printed_header = False
    if config.sections():
        for section_name in config.sections():
            if not printed_header and int(os.environ.get('RANK', '0')) == 0:
                print(config[section_name].get('header'))
                printed_header = True
            stats = run_experiment(config[section_name])
            if parallel.rank() == 0:
                print(section_name + ", " + str(stats['mean']) + ", " + str(stats['std']))
    else:
        val = run_experiment(config[config.default_section])
        print(val)


if __name__ == "__main__":
    main()
