#!/bin/bash
# bench.py --dagger: the round's updates on dense slots (mgp_replay_gather_many / _rows + mgp_train_step_indexed: MGP_FRAME_AGG=0)
# against aggregated slots (mgp_replay_aggregate + mgp_train_step_agg, the default).  -> profiles/r04_dagger_update_slots.txt
for agg in 0 1; do
for cfg in "--steps 500 --warmup 20" "--episodes 64 --agents 1000 --steps 200 --warmup 10 --updates 128" "--episodes 256 --agents 300 --steps 200 --warmup 10 --updates 256" "--episodes 256 --agents 200 --taps 4 --steps 200 --warmup 10 --updates 512"; do
  MGP_FRAME_AGG=$agg python bench.py --dagger $cfg 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); u = d['updates']
print('MGP_FRAME_AGG=$agg N=%d K=%d: %s slots, %.1f us per update (%.0f updates/s), mean loss %.5f' % (d['config']['agents'], d['config']['taps'], u['slots'], 1e3 * u['ms_per_update'], u['updates_per_s'], u['mean_loss']))"
done; done
