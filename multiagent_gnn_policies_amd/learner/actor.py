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
        """n_s features and n_a action axes per agent; `hidden_layers` lists the hidden widths; `k` delay taps enter
        the filter of layer `ind_agg`, in front of which the one aggregation takes place."""
        super().__init__()
        self.k, self.n_s, self.n_a, self.ind_agg = k, n_s, n_a, ind_agg
        self.layers = [n_s, *hidden_layers, n_a]
        self.n_layers = len(self.layers) - 1
        # parameter containers: shapes, names and creation order (hence RNG consumption) of reference actor.py:30-42
        convs = []
        for i, (c_in, c_out) in enumerate(zip(self.layers[:-1], self.layers[1:])):
            taps = k if i == ind_agg else 1
            convs.append(nn.Conv2d(c_in, c_out, kernel_size=(taps, 1), stride=(taps, 1)))
        self.conv_layers = nn.ModuleList(convs)
        self.use_fused = True       # fused single-kernel forward when the shape is covered (ind_agg == 0)

    def _check_shapes(self, delay_state, delay_gso):
        """AssertionError on any mismatch, as the reference's asserts (actor.py:53-61)."""
        B, K, F, N = delay_state.shape
        assert tuple(delay_gso.shape) == (B, self.k, N, N) and (K, F) == (self.k, self.n_s), \
            "expected delay_state (B,%d,%d,N) and delay_gso (B,%d,N,N), got %s and %s" % (
                self.k, self.n_s, self.k, tuple(delay_state.shape), tuple(delay_gso.shape))
        return B, N

    def forward(self, delay_state, delay_gso):
        """delay_state (B,K,F,N): features x_t, x_{t-1}, ...; delay_gso (B,K,N,N): delayed graph-shift operators
        I, A_t, A_t A_{t-1}, ...  ->  (B,1,nA,N)."""
        batch_size, n_agents = self._check_shapes(delay_state, delay_gso)

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
