"""`Actor` -- drop-in for the reference policy network (reference learner/actor.py:7-86).

Same constructor, attributes, `forward(delay_state, delay_gso)` contract, assertion behaviour and
`state_dict` layout (`conv_layers.{i}.weight (out,in,step,1)` / `.bias`), so the shipped checkpoint
loads and default initialisation consumes the torch RNG exactly like the reference.  The arithmetic is
NOT ATen: aggregation, filter GEMM, tanh MLP readout and their backward are the HIP kernels of
libmgp.so (ops.py).  `nn.Conv2d` modules are kept purely as parameter containers.
"""
import torch.nn as nn

from .. import ops


class Actor(nn.Module):

    def __init__(self, n_s, n_a, hidden_layers, k, ind_agg):
        """
        :param n_s: number of MDP states (features) per agent
        :param n_a: number of MDP actions per agent
        :param hidden_layers: list of hidden-layer widths
        :param k: aggregation filter length (number of delay taps)
        :param ind_agg: index of the layer before which the single aggregation happens
        """
        super(Actor, self).__init__()
        self.k = k
        self.n_s = n_s
        self.n_a = n_a
        self.layers = [n_s] + list(hidden_layers) + [n_a]
        self.n_layers = len(self.layers) - 1
        self.ind_agg = ind_agg
        # parameter containers with the reference's shapes and init order (actor.py:30-42)
        self.conv_layers = nn.ModuleList([
            nn.Conv2d(in_channels=self.layers[i], out_channels=self.layers[i + 1],
                      kernel_size=((k if i == ind_agg else 1), 1), stride=((k if i == ind_agg else 1), 1))
            for i in range(self.n_layers)])
        self.use_fused = True       # fused single-kernel forward when the shape is covered (ind_agg == 0)

    def forward(self, delay_state, delay_gso):
        """
        :param delay_state: (B,K,F,N) history of features x_t, x_{t-1}, ...
        :param delay_gso:   (B,K,N,N) delayed graph-shift operators I, A_t, A_t A_{t-1}, ...
        :return: (B,1,nA,N)
        """
        batch_size = delay_state.shape[0]
        n_agents = delay_state.shape[3]
        # same contract as reference actor.py:53-61
        assert delay_gso.shape[0] == batch_size
        assert delay_gso.shape[2] == n_agents
        assert delay_gso.shape[3] == n_agents
        assert delay_state.shape[1] == self.k
        assert delay_state.shape[2] == self.n_s
        assert delay_gso.shape[1] == self.k

        if self.use_fused and self.ind_agg == 0:
            from . import actor_fused
            out = actor_fused.try_forward(self, delay_state, delay_gso)
            if out is not None:
                return out

        x = delay_state.permute(0, 2, 1, 3)                    # (B,F,K,N) view, no copy
        for i in range(self.n_layers):
            if i == self.ind_agg:
                x = ops.aggregate(x, delay_gso)                # (B,C,K,N)
            conv = self.conv_layers[i]
            out_c, in_c, step, _ = conv.weight.shape
            B, C, T, N = x.shape
            if step > 1:
                assert T == step, "the (k,1) filter expects exactly k delay taps"
                x = x.reshape(B, C * T, 1, N)                  # rows (c,k) -- matches W.view(out, in*k)
            act = ops.ACT_TANH if i < self.n_layers - 1 else ops.ACT_NONE
            x = ops.dense(x, conv.weight.view(out_c, in_c * step), conv.bias, act)
        return x.view((batch_size, 1, self.n_a, n_agents))
