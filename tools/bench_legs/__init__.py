"""The legs of bench.py that are not the contract line itself: dense-kernel rooflines, parity gate, CPU baseline, the DAGGER
measurements, rank launching.  bench.py imports them; the emitted JSON is unchanged by the split."""
