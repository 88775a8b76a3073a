"""Test / benchmark infrastructure shared by tests/, bench.py, __graft_entry__.smoke() and the oracle's fixture generator.
NOT part of the product (diarizen_amd/ never imports it): seeded random-initialised state_dicts with the reference
checkpoints' key names (no pretrained weights exist offline), the seeded "turn-taking" weights + their calibration
(testkit/data/cal_*.npz, fitted by oracle/calibrate.py), and the synthetic recordings the benchmark runs on."""
