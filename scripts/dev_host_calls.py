import importlib, sys, json
sys.path.insert(0, ".")
import bench
abi = importlib.import_module("anticipated-vins-mono_amd.abi"); synth = importlib.import_module("anticipated-vins-mono_amd.synth")
import torch
r = bench.host_call_latency(abi, synth, reps=30)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("value", "min", "device_calls_ms", "selected_per_call")} for k, v in r.items()}))
