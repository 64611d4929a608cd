import json, sys, os
sys.path.insert(0, os.getcwd())
from tools.f16x2_bar_check import run_case, g19
for B, P, T in ((2, 3, 2), (1, 5, 5)):
    print(json.dumps(run_case(B, P, T, "bf16x3_r3", False)), flush=True)
print(json.dumps(g19("bf16x3_r3", False)), flush=True)
