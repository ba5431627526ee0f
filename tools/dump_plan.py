"""Write the tuned (tile, split_k) table of the 16-bit BASELINE configurations to one JSON file (GPU box):
    MILLIEYE_TUNE_CACHE=gpurun_out/plan_16bit_configs.json python tools/dump_plan.py
It runs exactly the forwards of tests/test_gpu_configs.py::test_module2_batch32_16bit / test_full_pipeline_608_batch16_fp32_and_f16
(fp32 + 16-bit, the batch run + the batch-1 runs), so every layer shape those tests plan is measured here, once, by the engine's
autotuner; the committed copy (tests/golden/plan_16bit_configs.json) pins the tests to the plan the benchmark runs instead
of the cold-start tiles - the same arithmetic on every GPU box."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
assert os.environ.get("MILLIEYE_TUNE_CACHE"), "set MILLIEYE_TUNE_CACHE to the file to write"
import torch  # noqa: E402

from millieye_amd import engine, synth  # noqa: E402
from tests import test_gpu_configs as tc  # noqa: E402


def main():
    with torch.no_grad():
        net = tc._m2_net("m2b32")
        net = net.to(net.device)
        x = torch.from_numpy(synth.uniform("m2b32/x", (32, 3, 416, 416))).cuda()
        for dtype in ("f32", "bf16", "f16"):
            net.base_detector.compute_dtype = dtype
            net(x)
            net(x[:1])
        del net, x
        torch.cuda.empty_cache()
        net = tc._net608("full608").cuda()
        x = torch.from_numpy(synth.uniform("full608/x", (16, 3, 608, 608))).cuda()
        maps, rboxes = synth.radar_inputs("full608/radar", 16, 608 // 16, boxes_per_image=2)
        maps, rboxes = torch.from_numpy(maps).cuda(), torch.from_numpy(rboxes)
        for dtype in ("f32", "f16"):
            net.base_detector.compute_dtype = dtype
            net(x, maps, rboxes.clone().cuda(), 0)
            rb = rboxes[rboxes[:, 0] == 0].clone()
            net(x[:1], maps[:1], rb.cuda(), 0)
    engine._tune_save()
    print("measured", engine._TUNE_STATS["measured"], "layer shapes ->", os.environ["MILLIEYE_TUNE_CACHE"], len(engine._TUNE_CACHE), "entries")


if __name__ == "__main__":
    main()
