"""Experiment runner with the reference's command line, cfg format and printed output (reference train.py:15-67):

    python3 train.py cfg/dagger.cfg

Every section of the INI file is one experiment (inheriting `[DEFAULT]`); the `header` key is printed once, then one
line `section, mean, std` per experiment.  Environments come from this package's registry instead of gym / gym_flock and
all learning runs on the MI355X kernels.  Under torchrun (one process per GPU) episodes are sharded over the ranks and
rank 0 prints.  Extensions: `alg = dagger_vec` (device-resident vectorised DAGGER, `n_envs` episodes per GPU);
`python3 train.py <cfg> --jobs J` runs J sections of the file side by side (same output, file order).
"""
import configparser
import os
import random
import sys

import numpy as np
import torch

from multiagent_gnn_policies_amd import envs, parallel


def seed_everything(seed, env):
    """The reference seeds four streams with one number (train.py:24-28): env, random, numpy, torch."""
    env.seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def _dagger(env, args, device):
    from multiagent_gnn_policies_amd.learner.gnn_dagger import train_dagger
    return train_dagger(env, args, device)


def _cloning(env, args, device):
    from multiagent_gnn_policies_amd.learner.gnn_cloning import train_cloning
    return train_cloning(env, args, device)


def _baseline(env, args, device):
    from multiagent_gnn_policies_amd.learner.gnn_baseline import train_baseline
    return train_baseline(env, args)


def _dagger_vec(env, args, device):
    from multiagent_gnn_policies_amd.learner.vec_dagger import train_dagger_vec
    env.close()
    return train_dagger_vec(args, device, n_envs=args.getint('n_envs', fallback=64))


ALGORITHMS = {'dagger': _dagger, 'cloning': _cloning, 'baseline': _baseline, 'dagger_vec': _dagger_vec}


def check_experiment(args):
    """Host-only pre-flight of one cfg section: everything the run will read (reference train.py:17-42,
    gnn_dagger.py:29-50,128-146) must be present and well-typed, the env id and algorithm known.  Returns the
    FlockParams the env will use.  Raises before any device work, naming the section's problem."""
    alg = (args.get('alg') or '').lower()
    if alg not in ALGORITHMS:
        raise Exception('Invalid algorithm/mode name')
    env = envs.make(args.get('env'))                    # KeyError for ids outside the registry (e.g. the AirSim backend)
    env.env.params_from_cfg(args)
    args.getint('seed')
    if alg == 'baseline':
        args.getboolean('centralized'); args.getint('n_test_episodes')
    else:
        for key in ('n_states', 'n_actions', 'k', 'hidden_size', 'n_agents', 'batch_size', 'buffer_size',
                    'updates_per_step', 'n_train_episodes', 'test_interval', 'n_test_episodes'):
            if args.getint(key) is None:
                raise KeyError("cfg key %r missing" % key)
        for key in ('actor_lr', 'beta_coeff', 'gamma', 'tau'):
            if args.getfloat(key) is None:
                raise KeyError("cfg key %r missing" % key)
        args.getboolean('debug')
        if args.getint('n_states') != env.env.n_features or args.getint('n_actions') != env.env.nu:
            raise ValueError("n_states / n_actions do not match the environment (6 features, 2 action axes)")
    return env.env.params


def run_experiment(args):
    check_experiment(args)
    if not torch.cuda.is_available():
        raise RuntimeError("train.py needs an MI355X (HIP device); this framework has no CPU compute path")
    rank, _world, local_rank = parallel.init_from_env()
    device = torch.device("cuda", parallel.local_device_index(local_rank))
    torch.cuda.set_device(device)

    env = envs.make(args.get('env'), device=str(device))
    if isinstance(env.env, envs.FlockingRelativeEnv):
        env.env.params_from_cfg(args)
    seed_everything(args.getint('seed') + rank, env)      # + rank: every rank rolls out different episodes

    alg = args.get('alg').lower()
    if alg not in ALGORITHMS:
        raise Exception('Invalid algorithm/mode name')
    return ALGORITHMS[alg](env, args, device)


def iter_experiments(config):
    """(name, section) for every experiment of the file; a file without sections is one unnamed experiment."""
    names = config.sections()
    if not names:
        yield None, config[config.default_section]
    for name in names:
        yield name, config[name]


def run_sections_concurrently(path, names, jobs):
    """Extension (`python3 train.py <cfg> --jobs J`): the experiments of a cfg file are independent (every section seeds its own
    streams, reference train.py:24-28), so on a node with several GPUs J sections run side by side, one worker process each
    (`--section NAME`), dealt round-robin to the visible devices (LOCAL_RANK), and their result lines are printed in file order
    exactly as the sequential run prints them.  NOT a gain on ONE GPU: the four full trainings of
    cfg/flocking_dagger_vec_k_sweep.cfg take 6.8 s in one process and 15.0 s as four processes sharing the MI355X (each
    worker pays the framework start-up, and chains of short dependent launches from different processes interleave badly)."""
    import subprocess
    import tempfile
    import time
    results, running, order = {}, {}, list(names)
    pending = list(enumerate(order))
    try:                                                          # workers are dealt to the VISIBLE devices, not to `jobs` slots
        import torch
        n_dev = max(1, torch.cuda.device_count())
    except Exception:
        n_dev = 1
    try:
        while pending or running:
            while pending and len(running) < jobs:
                i, name = pending.pop(0)
                env = dict(os.environ, LOCAL_RANK=str(i % n_dev), MGP_TRAIN_WORKER='1')
                fo, fe = tempfile.TemporaryFile('w+'), tempfile.TemporaryFile('w+')   # (files, not pipes: a debug run prints a lot)
                proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), path, '--section', name], env=env,
                                        stdout=fo, stderr=fe, text=True)
                running[name] = (proc, fo, fe)
            finished = [n for n, (pr, _, _) in running.items() if pr.poll() is not None]
            if not finished:
                time.sleep(0.02)
                continue
            for name in finished:
                proc, fo, fe = running.pop(name)
                fo.seek(0); fe.seek(0)
                out, err = fo.read(), fe.read()
                fo.close(); fe.close()
                if proc.returncode != 0:
                    raise RuntimeError("section %r failed:\n%s" % (name, err[-4000:]))
                lines = [ln for ln in out.splitlines() if ln.startswith(name + ", ")]
                if not lines:                                     # never pass some other output line off as the result
                    raise RuntimeError("section %r printed no '%s, ...' result line; its output ended with:\n%s" % (name, name, out[-2000:]))
                results[name] = lines[-1]
                extra = [ln for ln in out.splitlines() if ln != results[name]]
                if extra:                                         # a section's own progress lines (debug = True), kept together
                    sys.stdout.write("\n".join(extra) + "\n")
                sys.stderr.write(err)
    finally:
        for proc, fo, fe in running.values():
            proc.kill()
            fo.close(); fe.close()
    return [results[n] for n in order]


def main(argv=None):
    argv = sys.argv if argv is None else argv
    fname = argv[1]
    here = os.path.dirname(os.path.abspath(__file__))
    path = fname if os.path.exists(fname) else os.path.join(here, fname)
    config = configparser.ConfigParser()
    config.read(path)
    is_root = int(os.environ.get('RANK', '0')) == 0
    header_done = False
    only = argv[argv.index('--section') + 1] if '--section' in argv else None
    jobs = int(argv[argv.index('--jobs') + 1]) if '--jobs' in argv else 1
    if jobs > 1 and only is None and config.sections() and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        names = config.sections()
        print(config[names[0]].get('header'))
        for line in run_sections_concurrently(path, names, min(jobs, len(names))):
            print(line)
        return
    for name, section in iter_experiments(config):
        if only is not None:
            if name != only:
                continue
            header_done = True                               # a worker prints its result line only
        if name is not None and not header_done and is_root:
            print(section.get('header'))
            header_done = True
        stats = run_experiment(section)
        if not is_root:
            continue
        if name is None:
            print(stats)
        else:
            print(name + ", " + str(stats['mean']) + ", " + str(stats['std']))


if __name__ == "__main__":
    main()
