"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain numpy (and one torch-CPU port used only as the
timed CPU baseline), the arithmetic of the reference's hot path:

  * actor.py        -- Actor forward / backward   (reference learner/actor.py:45-86)
  * state.py        -- delayed-GSO / delay line   (reference learner/state_with_delay.py:38-53)
  * dagger.py       -- MSE loss + Adam update     (reference learner/gnn_dagger.py:76-96)
  * flock.py        -- flocking sim step, reset, expert controller (gym_flock, NOT in
                       /root/reference: PARITY UNPINNED, follows this repo's own spec,
                       DESIGN.md section "FLOCK-SPEC v1")
  * synth.py        -- seeded synthetic (S, X) generators shared by tests / bench / goldens
  * torch_port.py   -- PyTorch-CPU restatement of the same op sequence (cpu_baseline leg)

Pinning: actor/state/dagger are checked against golden vectors produced by importing
the reference itself in the build container (tools/gen_golden.py -> tests/golden/*.npz).
flock.py has no reference to pin against ("parity unpinned").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package. The product (multiagent_gnn_policies_amd) never does.
"""
