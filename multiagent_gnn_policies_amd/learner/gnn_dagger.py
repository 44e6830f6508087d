"""DAGGER learner and training loop -- drop-in for reference learner/gnn_dagger.py:18-243.

`DAGGER(device, args, k=None)` exposes the same `select_action`, `gradient_step`, `save_model`,
`load_model`; `train_dagger(env, args, device)` keeps the reference's schedule (beta decay per episode,
expert labels, `updates_per_step` updates after each episode, periodic evaluation, final statistics).
Arithmetic (Actor forward/backward, MSE, Adam) runs in the HIP kernels; parameters and gradients live in
one flat fp32 buffer each, so a data-parallel run needs exactly one all-reduce of 6,920 bytes per update.
"""
import os

import numpy as np
import torch

from .. import ops
from .. import parallel
from ..parallel import FlatGradSync
from .actor import Actor


class FlatAdam(object):
    """torch.optim.Adam(defaults) semantics on a single flat parameter buffer (kernel mgp_adam_step).

    Re-homes every parameter of `module` as a view into one contiguous fp32 buffer (state_dict keys and
    shapes are unchanged), keeps a matching flat gradient buffer, first/second-moment buffers and the
    step count.
    """

    def __init__(self, module, lr, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in module.parameters()]
        self.lr, self.betas, self.eps = lr, betas, eps
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty((total,), device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                off += n
        self.flat_grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        # ONE step counter, on the device, shared by every update path (eager, HIP-graph replays of any batch size, the
        # indexed rounds of vec_dagger): each Adam kernel reads it for the bias corrections and advances it itself, so
        # mixing paths can never replay a stale step.  `step_count` mirrors it on the host (bookkeeping only).
        self.step_dev = torch.zeros((1,), device=dev, dtype=torch.int32)
        self.step_count = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """Pack p.grad into the flat gradient buffer (one device-side concat of 1,730 floats)."""
        torch.cat([p.grad.reshape(-1) for p in self.params], out=self.flat_grad)
        return self.flat_grad

    def step(self):
        ops.adam_step_dev(self.flat, self.flat_grad, self.m, self.v, self.lr, self.step_dev,
                          self.betas[0], self.betas[1], self.eps)
        self.step_count += 1

    def views(self, flat):
        """Per-parameter views into a flat buffer laid out like self.flat."""
        out, off = [], 0
        for p in self.params:
            out.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return out


class GraphedUpdate(object):
    """One DAGGER update captured once per batch size as a HIP graph on static buffers and replayed with one host call
    instead of ~40 Python-level ops.  Two launches when mgp_train_step covers the shape (forward + MSE gradient +
    backward per 16-column tile, then partial reduction + device-step Adam); otherwise five (fused forward with saved
    activations -> MSE gradient -> fused backward straight into the flat gradient buffer -> device-step Adam).

    Data-parallel runs (torch.distributed initialised, world > 1) use the same object: gradients into the flat buffer
    (mgp_train_grads: two launches, or forward / MSE / backward), ONE in-place all-reduce of that 6,920-byte buffer, the
    1/world scale, device-step Adam.  Under RCCL ("nccl") the whole sequence INCLUDING the collective is captured into the
    graph, so an update is still one host call; where the collective cannot be captured (gloo, or a runtime that refuses)
    the same pre-bound sequence is enqueued eagerly: four kernel launches + one collective, no tensor construction, no
    autograd, no host synchronisation."""

    def __init__(self, learner, B):
        import ctypes
        from .. import _lib
        from .actor_fused import _ptr_array
        actor, opt = learner.actor, learner.actor_optim
        dev = opt.flat.device
        K, N, F = actor.k, learner.n_agents, actor.n_s
        dims = tuple(actor.layers)
        self.cdims = (ctypes.c_int * len(dims))(*dims)
        L = _lib.lib()
        self.X = torch.empty((B, K, F, N), device=dev)
        self.G = torch.empty((B, K, N, N), device=dev)
        self.Y = torch.empty((B, 1, actor.n_a, N), device=dev)
        self.out = torch.empty((B, 1, actor.n_a, N), device=dev)
        self.dOut = torch.empty_like(self.out)
        self.loss = torch.zeros((1,), device=dev)
        self.saved = torch.empty((L.mgp_actor_saved_floats(self.cdims, actor.n_layers, B, K, N),), device=dev)
        self.ws = torch.empty((max(1, L.mgp_actor_bwd_workspace(self.cdims, actor.n_layers, B, K, N)),), device=dev)
        self.step_dev = opt.step_dev                # shared with every other update path (FlatAdam)
        pv, gv = opt.views(opt.flat), opt.views(opt.flat_grad)
        self.Wp, self.bp = _ptr_array(pv[0::2]), _ptr_array(pv[1::2])
        self.dWp, self.dbp = _ptr_array(gv[0::2]), _ptr_array(gv[1::2])
        self._keep = (pv, gv)
        self.B, self.K, self.N, self.nl = B, K, N, actor.n_layers
        self.opt = opt
        self.graph = None
        self.dist = parallel.is_distributed()
        self.world = parallel.world_size()
        fused_train = bool(learner.use_train_step and L.mgp_train_supported(self.cdims, actor.n_layers, B, K, N))
        # data parallel with the one-shot exchange up: still two launches, the exchange runs inside the second one
        self.p2p = learner.p2p if (self.dist and fused_train and getattr(learner, 'p2p', None) is not None
                                   and learner.p2p.n_floats > opt.flat.numel()) else None
        self.two_launch = fused_train and not self.dist          # reduction + Adam fused: no room for a collective
        self.train_grads = fused_train and self.dist and self.p2p is None   # gradients only, Adam after the all-reduce
        if fused_train:
            self.tws = torch.zeros((L.mgp_train_workspace(self.cdims, actor.n_layers, B, K, N),), device=dev)
        # the collective is captured into the HIP graph under RCCL; gloo moves data through the host and cannot be
        self.capturable = (not self.dist) or self.p2p is not None or (torch.distributed.get_backend() == 'nccl'
                                                                      and os.environ.get('MGP_DIST_GRAPH', '1') != '0')

    def _enqueue(self):
        from .. import _lib
        L, o = _lib.lib(), self.opt
        st = ops._stream()
        if self.two_launch:
            _lib.check(L.mgp_train_step(ops._ptr(self.X), ops._ptr(self.G), ops._ptr(self.Y), ops._ptr(o.flat),
                                        ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v), self.cdims, self.nl,
                                        o.lr, o.betas[0], o.betas[1], o.eps, ops._ptr(self.step_dev),
                                        ops._ptr(self.loss), ops._ptr(self.tws), self.B, self.K, self.N, st),
                       'mgp_train_step')
            return
        if self.p2p is not None:
            _lib.check(L.mgp_train_step_p2p(ops._ptr(self.X), ops._ptr(self.G), ops._ptr(self.Y), None, None, None, 0,
                                            ops._ptr(o.flat), ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v), self.cdims,
                                            self.nl, o.lr, o.betas[0], o.betas[1], o.eps, ops._ptr(self.step_dev),
                                            ops._ptr(self.loss), ops._ptr(self.tws), self.B, self.K, self.N,
                                            self.p2p.handle, st), 'mgp_train_step_p2p')
            return
        if self.train_grads:
            _lib.check(L.mgp_train_grads(ops._ptr(self.X), ops._ptr(self.G), ops._ptr(self.Y), self.Wp, self.bp, self.cdims,
                                         self.nl, ops._ptr(o.flat_grad), ops._ptr(self.loss), ops._ptr(self.tws),
                                         self.B, self.K, self.N, st), 'mgp_train_grads')
        else:
            _lib.check(L.mgp_actor_fwd(ops._ptr(self.X), ops._ptr(self.G), self.Wp, self.bp, self.cdims, self.nl,
                                       ops._ptr(self.out), ops._ptr(self.saved), self.B, self.K, self.N, st), 'mgp_actor_fwd')
            _lib.check(L.mgp_mse_grad(ops._ptr(self.out), ops._ptr(self.Y), ops._ptr(self.dOut), ops._ptr(self.loss),
                                      self.out.numel(), st), 'mgp_mse_grad')
            _lib.check(L.mgp_actor_bwd(ops._ptr(self.dOut), ops._ptr(self.saved), self.Wp, self.cdims, self.nl, self.dWp,
                                       self.dbp, self.B, self.K, self.N, ops._ptr(self.ws), st), 'mgp_actor_bwd')
        if self.dist:
            # the ONE exchange of a data-parallel update: 1,730 floats, in place, summed then scaled (reference semantics
            # of a world-times larger minibatch); latency-bound, so never split per tensor
            torch.distributed.all_reduce(o.flat_grad, op=torch.distributed.ReduceOp.SUM)
            o.flat_grad.div_(self.world)                     # as FlatGradSync: bit-identical paths for any world size
        _lib.check(L.mgp_adam_step_dev(ops._ptr(o.flat), ops._ptr(o.flat_grad), ops._ptr(o.m), ops._ptr(o.v),
                                       o.flat.numel(), o.lr, o.betas[0], o.betas[1], o.eps, ops._ptr(self.step_dev), st),
                   'mgp_adam_step_dev')

    def run(self, X, G, Y):
        # callers that gathered their batch straight into the static buffers (DeviceReplay.sample(out=...)) skip the copies
        if X is not self.X:
            self.X.copy_(X)
        if G is not self.G:
            self.G.copy_(G)
        if Y is not self.Y:
            self.Y.copy_(Y)
        if self.graph is None and self.capturable:
            if self.dist and self.p2p is None:
                parallel.warm_up_collective(self.opt.flat.device)   # communicator set-up must not happen under capture
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with ops.graph_capture(graph):
                    self._enqueue()
                self.graph = graph
            except RuntimeError as e:                 # a collective the runtime refuses to capture: stay eager, loudly
                if not self.dist or self.p2p is not None:
                    raise
                import warnings
                warnings.warn("data-parallel update: HIP-graph capture of the all-reduce failed (%s); updates are enqueued "
                              "eagerly (same kernels, same collective)" % (e,), RuntimeWarning)
                self.capturable = False
                torch.cuda.synchronize()
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue()
        self.opt.step_count += 1
        return self.loss


class DAGGER(object):

    def __init__(self, device, args, k=None):
        n_s = args.getint('n_states')
        n_a = args.getint('n_actions')
        k = k or args.getint('k')
        hidden_size = args.getint('hidden_size')
        n_layers = args.getint('n_layers') or 2
        self.gamma = args.getfloat('gamma')      # read for cfg compatibility; unused by DAGGER
        self.tau = args.getfloat('tau')

        self.n_agents = args.getint('n_agents')
        self.n_states = n_s
        self.n_actions = n_a
        self.device = torch.device(device)

        hidden_layers = [hidden_size] * n_layers
        ind_agg = 0                               # reference gnn_dagger.py:43
        self.actor = Actor(n_s, n_a, hidden_layers, k, ind_agg).to(self.device)
        self.actor_optim = FlatAdam(self.actor, lr=args.getfloat('actor_lr'))
        self.grad_sync = FlatGradSync()           # no-op unless torch.distributed is initialised
        self.grad_sync.broadcast_(self.actor_optim.flat)
        # data-parallel runs: the one-shot exchange of the flat gradient (+ the loss), csrc/p2p_device.h; None = the
        # torch.distributed collective (single process, MGP_P2P=0, or the exchange could not be brought up)
        self.p2p = parallel.P2PExchange.create(self.actor_optim.flat.numel() + 1, self.device)
        self.grad_sync.p2p = self.p2p
        self._graphed = {}                        # batch size -> GraphedUpdate
        self.use_graphed_update = True
        self.use_train_step = True                # two-launch update (mgp_train_step / mgp_train_grads) when covered
        self._train_ws = {}                       # batch size -> workspace of the eager mgp_train_grads path

    def begin_updates(self):
        """Call before a round of updates of a data-parallel run: aligns the ranks on the host, so that no rank's exchange
        kernel polls for a peer that is still seconds away (the exchange gives up after 5 s)."""
        if self.p2p is not None:
            torch.distributed.barrier()
            # A timed-out exchange leaves the weights in a state no rank can trust: the fused reduce (csrc/train_step.hip,
            # train_reduce_kernel<P2P>) skips Adam for the entries whose own poll gave up and for workgroups that saw the
            # sticky status word, but entries that had arrived in time were stepped -- differently on each rank.  The round's
            # starting point (weights, both moments, step counter: 21 KB) is kept, and end_updates() rolls back to it before
            # it raises.
            o = self.actor_optim
            self._round_start = (o.flat.clone(), o.m.clone(), o.v.clone(), o.step_dev.clone(), o.step_count)

    def end_updates(self):
        """After a round of updates: if an exchange timed out ON ANY RANK (synchronises the stream; one MAX all-reduce of the
        status flag), every rank rolls the weights, the Adam moments and the step counter back to where begin_updates() found
        them and raises.  The agreement is collective because the failure is not: the rank whose poll gave up skipped Adam for
        the entries it missed, while a peer that merely arrived late found every packet waiting and stepped -- a caller that
        caught the error on one rank and went on would train on diverged ranks.  After a failure the exchange's sequence
        numbers are out of step: call reset_exchange() (collective) before the next round."""
        if self.p2p is None:
            return
        snap, self._round_start = getattr(self, '_round_start', None), None
        st, _ = self.p2p.status()
        if parallel.any_rank(st != 0):
            if snap is not None:
                o = self.actor_optim
                o.flat.copy_(snap[0]); o.m.copy_(snap[1]); o.v.copy_(snap[2]); o.step_dev.copy_(snap[3])
                o.step_count = snap[4]
            from .._lib import MgpError
            raise MgpError("one-shot gradient exchange: a peer did not publish within the timeout on %s (rank %d of %d)%s"
                           % ("this rank" if st != 0 else "another rank", self.p2p.rank, self.p2p.world,
                              "; weights, moments and step counter restored to the start of the round" if snap is not None else ""))

    def reset_exchange(self):
        """Collective: tears the one-shot exchange down and brings a fresh one up (new mailboxes, sequence numbers at zero).
        The only way on after end_updates() raised.  Update graphs captured with the old exchange are dropped (rebuilt on the
        next update); FrameUpdates objects bound to this learner must be rebuilt by their owner."""
        if self.p2p is not None:
            torch.cuda.synchronize()
            self.p2p.close()
        self._graphed = {}
        self.p2p = parallel.P2PExchange.create(self.actor_optim.flat.numel() + 1, self.device)
        self.grad_sync.p2p = self.p2p
        return self.p2p is not None

    def _checked_step(self):
        """Adam after an eager all-reduce.  When that all-reduce was the one-shot exchange (stand-alone kernel,
        FlatGradSync), its status is read BEFORE the step is enqueued: a late peer's share counted as zero must never
        reach the weights (costs one stream synchronisation per update; the graph paths check inside the kernel)."""
        if self.p2p is not None and parallel.is_distributed():
            self.p2p.check()
        self.actor_optim.step()

    def __del__(self):
        try:
            if getattr(self, 'p2p', None) is not None:
                self.p2p.close()
        except Exception:
            pass

    def _can_graph(self, X):
        """Runs whose shape the fused kernels cover go through GraphedUpdate: a HIP-graph replay per update (the
        data-parallel all-reduce included, under RCCL), or the same pre-bound launch sequence enqueued eagerly (gloo)."""
        import ctypes
        from .. import _lib
        if not self.use_graphed_update or not self.actor.use_fused:
            return False
        dims = tuple(self.actor.layers)
        cd = (ctypes.c_int * len(dims))(*dims)
        return bool(_lib.lib().mgp_actor_supported(cd, self.actor.n_layers, self.actor.k, X.shape[3])) and X.is_cuda

    def select_action(self, state):
        """(1,K,F,N),(1,K,N,N) -> action (N,nA) on the device (reference gnn_dagger.py:55-72)."""
        # (the reference brackets this with actor.eval() / actor.train(); the Actor has no mode-dependent layer, and the
        #  two recursive flag updates cost 40 us per environment step here, so the flags are left alone)
        with torch.no_grad():
            mu = self.actor(state.delay_state, state.delay_gso)
        return mu.permute(0, 1, 3, 2).reshape((self.n_agents, self.n_actions))

    def gradient_step(self, batch):
        """One supervised update on a batch of transitions (reference gnn_dagger.py:76-96)."""
        gsos = tuple(s.delay_gso for s in batch.state)
        states = tuple(s.delay_state for s in batch.state)
        # concatenate straight into the HIP-graph update's static input buffers when that path will run (saves three
        # device copies per update); otherwise into fresh tensors as the reference does
        bufs = None
        if states[0].device == self.device and all(t.shape[0] == 1 for t in states):
            bufs = self.graphed_buffers(len(states), states[0].shape[3])
        if bufs is not None:
            X, G, Y = bufs
            torch.cat(states, out=X); torch.cat(gsos, out=G); torch.cat(batch.action, out=Y)
            return self.gradient_step_tensors(X, G, Y)
        delay_gso_batch = torch.cat(gsos).to(self.device)
        delay_state_batch = torch.cat(states).to(self.device)
        optimal_action_batch = torch.cat(batch.action).to(self.device)
        return self.gradient_step_tensors(delay_state_batch, delay_gso_batch, optimal_action_batch)

    def graphed_buffers(self, B, N):
        """(X, G, Y) static input buffers of the HIP-graph update for batch size B, or None when updates run eagerly
        (distributed run / shape outside the fused kernels): gather a batch straight into them to skip three copies."""
        probe = torch.empty((0, self.actor.k, self.n_states, N), device=self.device)
        if not self._can_graph(probe):
            return None
        gu = self._graphed.get(B)
        if gu is None:
            gu = self._graphed[B] = GraphedUpdate(self, B)
        return gu.X, gu.G, gu.Y

    def gradient_step_tensors(self, delay_state_batch, delay_gso_batch, optimal_action_batch, sync=True):
        """One update on device tensors.  sync=False returns the loss as a (1,) device tensor (graph path only) so the
        host can queue the next update without waiting for this one."""
        if self._can_graph(delay_state_batch):
            B = delay_state_batch.shape[0]
            gu = self._graphed.get(B)
            if gu is None:
                gu = self._graphed[B] = GraphedUpdate(self, B)
            loss = gu.run(delay_state_batch, delay_gso_batch, optimal_action_batch)
            if not sync:
                return loss.clone()
            value = loss.item()
            # the stream is idle anyway: outside a begin_updates() / end_updates() round a timed-out exchange raises HERE, per
            # update (rank-local); inside a round the status word is sticky and end_updates() settles it on every rank together
            if self.p2p is not None and getattr(self, '_round_start', None) is None:
                self.p2p.check()
            return value
        loss = self._train_grads(delay_state_batch, delay_gso_batch, optimal_action_batch)
        if loss is not None:                       # two launches wrote the flat gradient; (all-reduce,) Adam
            self.grad_sync.all_reduce_mean_(self.actor_optim.flat_grad)
            self._checked_step()
            return loss.item() if sync else loss
        self.actor_optim.zero_grad()
        actor_batch = self.actor(delay_state_batch, delay_gso_batch)
        policy_loss = ops.mse_loss(actor_batch, optimal_action_batch)
        policy_loss.backward()
        flat_grad = self.actor_optim.gather_grads()
        self.grad_sync.all_reduce_mean_(flat_grad)
        self._checked_step()
        return policy_loss.item()

    def _train_grads(self, X, G, Y):
        """mgp_train_grads: forward + MSE + parameter backward into the flat gradient buffer; returns the (1,) loss
        tensor, or None when the shape is outside its coverage (the caller composes the separate kernels)."""
        import ctypes
        from .. import _lib
        from .actor_fused import _ptr_array
        actor, opt = self.actor, self.actor_optim
        if not (self.use_train_step and actor.use_fused and actor.ind_agg == 0 and X.is_cuda):
            return None
        B, K, _, N = X.shape
        dims = tuple(actor.layers)
        cd = (ctypes.c_int * len(dims))(*dims)
        L = _lib.lib()
        if not L.mgp_train_supported(cd, actor.n_layers, B, K, N):
            return None
        ws = self._train_ws.get((B, N))
        if ws is None:
            ws = self._train_ws[(B, N)] = torch.zeros((L.mgp_train_workspace(cd, actor.n_layers, B, K, N),),
                                                      device=X.device)
        pv = opt.views(opt.flat)
        loss = torch.empty((1,), device=X.device)
        X, G, Y = X.contiguous(), G.contiguous(), Y.contiguous()
        _lib.check(L.mgp_train_grads(ops._ptr(X), ops._ptr(G), ops._ptr(Y), _ptr_array(pv[0::2]), _ptr_array(pv[1::2]),
                                     cd, actor.n_layers, ops._ptr(opt.flat_grad), ops._ptr(loss), ops._ptr(ws),
                                     B, K, N, ops._stream()), 'mgp_train_grads')
        return loss

    def save_model(self, env_name, suffix="", actor_path=None):
        if not os.path.exists('models/'):
            os.makedirs('models/')
        if actor_path is None:
            actor_path = "models/actor_{}_{}".format(env_name, suffix)
        print('Saving model to {}'.format(actor_path))
        torch.save({k: v.detach().clone() for k, v in self.actor.state_dict().items()}, actor_path)

    def load_model(self, actor_path, map_location):
        if actor_path is not None:
            if str(actor_path).endswith('.npz'):   # weight fixture: keys with '.' spelled '__' (tests/golden/gen_golden.py)
                import numpy as np
                with np.load(actor_path) as z:
                    sd = {k_.replace('__', '.'): torch.from_numpy(z[k_]) for k_ in z.files if k_ != 'meta'}   # ('meta': how a
                    #                                                  policy of tests/golden/policies was trained -- a JSON string)
            else:
                sd = torch.load(actor_path, map_location)
            own = self.actor.state_dict()
            # validate everything before touching the live parameters (nn.Module.load_state_dict semantics: reference
            # gnn_dagger.py:122 raises on unexpected / missing keys and on shape mismatches, leaving the model unchanged)
            unexpected, missing = sorted(set(sd) - set(own)), sorted(set(own) - set(sd))
            if unexpected or missing:
                raise KeyError("state_dict mismatch: unexpected keys %s, missing keys %s" % (unexpected, missing))
            bad = ["%s: checkpoint %s vs model %s" % (k_, tuple(sd[k_].shape), tuple(own[k_].shape))
                   for k_ in own if tuple(sd[k_].shape) != tuple(own[k_].shape)]
            if bad:
                raise RuntimeError("state_dict shape mismatch: " + "; ".join(bad))
            with torch.no_grad():
                for k_, v in sd.items():
                    own[k_].copy_(v)              # in place: keeps the flat-buffer views intact


class BetaSchedule(object):
    """beta of global episode e, computed exactly as the reference does (gnn_dagger.py:141,148): `beta = 1` before the
    loop, then `beta = max(beta * beta_coeff, 0.5)` once per episode -- a RUNNING product, bit for bit (a closed form
    beta_coeff ** (e + 1) differs in the last ulp, and beta is the probability handed to np.random.binomial).  Ranks of a
    data-parallel run ask for non-consecutive episodes, so earlier values are memoised."""

    def __init__(self, beta_coeff):
        self.beta_coeff = beta_coeff
        self._betas = []

    def __call__(self, episode):
        while len(self._betas) <= episode:
            prev = self._betas[-1] if self._betas else 1
            self._betas.append(max(prev * self.beta_coeff, 0.5))
        return self._betas[episode]


def train_dagger(env, args, device):
    """Reference gnn_dagger.py:126-243.  Evaluation while training only when `debug`; the statistics are those of a final
    evaluation, after which the model is saved (`debug` and `fname`)."""
    from .imitation import ImitationRun
    run = ImitationRun(env, DAGGER(device, args), args, device)
    return run.run(BetaSchedule(args.getfloat('beta_coeff')), eval_always=False, keep_best=False)
