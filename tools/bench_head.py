"""Classification-feature head timing (features.py:49-73): Conv2d(1024, Cout, 3, padding=1, bias=False) + InstanceL2Norm on
18x18 layer-3 maps -- DiMP-50: Cout = 512, one frame; ToMP: Cout = 256, test frame + two memory frames per tracked frame.
    python tools/bench_head.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracking_amd import _lib  # noqa: E402
from pytracking_amd import features as FM  # noqa: E402


def main():
    if _lib.needs_build():
        _lib.build_library()
    dev = torch.device("cuda", 0)
    for name, n, cout in (("dimp50", 1, 512), ("tomp", 3, 256)):
        head = FM.residual_bottleneck(feature_dim=256, num_blocks=0, l2norm=True, final_conv=True,
                                      norm_scale=(1.0 / (cout * 16)) ** 0.5, out_dim=cout).to(dev).eval()
        x = torch.randn(n, 1024, 18, 18, device=dev)
        ref = torch.nn.Sequential(head[0])                                  # stock MIOpen convolution for comparison
        with torch.no_grad():
            for _ in range(5):
                head(x)
                ref(x)
            out = {}
            for tag, fn in (("fused_us", head), ("stock_conv_only_us", ref)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    fn(x)
                torch.cuda.synchronize()
                out[tag] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
        fl = 2.0 * n * 324 * 1024 * 9 * cout
        out.update(workload=f"clf head {name}: {n} x 1024x18x18 -> {cout}", GFLOP=round(fl / 1e9, 2),
                   TFLOPs=round(fl / out["fused_us"] / 1e6, 1))
        print(json.dumps(out))


if __name__ == "__main__":
    main()
