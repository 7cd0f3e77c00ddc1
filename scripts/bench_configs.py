"""fps of the BASELINE.json configs that fit this engine (1: 256^2 1-step, 2: 512^2 4-step, 5: 768^2 8-step + ControlNet,
plus the UI's live 768x432 4-step), one frame per launch: a frame alone (ControlNet encoder on the lane's side stream) and four
launch lanes busy.  Reporting only; bench.py is the contract."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps
ops = HipOps(0)
ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
slots = [eng, eng.make_slot(), eng.make_slot(), eng.make_slot()]
out = []
for name, H, Wd, steps, cn, scale in [("config1 256x256 1-step +CN", 256, 256, 1, True, 1.0), ("config2 512x512 4-step +CN", 512, 512, 4, True, 1.0),
                                       ("config2 512x512 4-step no CN", 512, 512, 4, False, 1.0), ("config5 768x768 8-step +CN scale 2", 768, 768, 8, True, 2.0),
                                       ("UI live 768x432 4-step +CN scale 2", 432, 768, 4, True, 2.0)]:
    f = np.random.default_rng(0).integers(0, 256, (H, Wd, 3), dtype=np.uint8)
    eng.overlap_launch = True
    eng.prepare(H, Wd, steps, 0.6, controlnet_scale=scale, use_controlnet=cn)
    lat = []
    for i in range(12):
        t = time.perf_counter(); eng.infer_u8(f); lat.append((time.perf_counter() - t) * 1e3)
    for e in slots:
        e.overlap_launch = False
        e.prepare(H, Wd, steps, 0.6, controlnet_scale=scale, use_controlnet=cn)
        e.ops.upload(e.frame_u8, torch.from_numpy(f))
    for i in range(8): slots[i % 4].launch()
    for e in slots: e.ops.synchronize()
    n = 48
    t = time.perf_counter()
    for i in range(n): slots[i % 4].launch()
    for e in slots: e.ops.synchronize()
    fps4 = n / (time.perf_counter() - t)
    # (arena: the activation memory ONE engine -- one program x frames per launch x lane -- owns; a worker's plan cache holds
    #  lanes x batch sizes x programs of them under `memory_budget`)
    row = {"config": name, "p50_latency_ms_1_in_flight": round(sorted(lat)[6], 2), "fps_4_lanes": round(fps4, 1),
           "arena_gb_per_engine_one_frame_per_launch": round(eng.plan["arena_bytes"] / 2 ** 30, 2)}
    out.append(row)
    print(json.dumps(row), flush=True)

# ---- the reference-only mode (lcm_reference_pipeline.py:855-890: a WRITE pass and a READ pass of the UNet per step,
#      no ControlNet, one frame per launch): what `ref=True` costs next to the plain UNet-only frame above
eng.overlap_launch = True
eng.prepare(512, 512, 4, 0.6, use_controlnet=False, ref_mode=True)
f = np.random.default_rng(0).integers(0, 256, (512, 512, 3), dtype=np.uint8)
eng.ops.upload(eng.ref_u8, torch.from_numpy(np.random.default_rng(1).integers(0, 256, (512, 512, 3), dtype=np.uint8)))
lat = []
for i in range(12):
    t = time.perf_counter(); eng.infer_u8(f); lat.append((time.perf_counter() - t) * 1e3)
row = {"config": "reference-only (ref=True) 512x512 4-step, no CN, 1 frame per launch", "p50_latency_ms_1_in_flight": round(sorted(lat)[6], 2),
       "fps_1_in_flight": round(1e3 / sorted(lat)[6], 1)}
out.append(row)
print(json.dumps(row), flush=True)

# ---- BASELINE configs[3]: SDXL 1024x1024 LCM 4-step (UNet only path: no ControlNet), 1 frame per launch
del slots, eng, wu, wc
torch.cuda.empty_cache()
wx = W.synthesize(W.unet_spec(C.SDXL_UNET), "sdxl.", device="cuda")
xl = Engine(ops, C.SDXL_UNET, None, C.TAESD, wx, None, wv)
g = torch.Generator().manual_seed(11)
xl.set_text_embeds((torch.randn(77, 2048, generator=g) * 0.5).half())
xl.set_added_cond((torch.randn(1280, generator=g) * 0.5).half(), (1024, 1024, 0, 0, 1024, 1024))
xl.prepare(1024, 1024, 4, 0.6, use_controlnet=False)
f = np.random.default_rng(0).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
lat = []
for i in range(8):
    t = time.perf_counter(); xl.infer_u8(f); lat.append((time.perf_counter() - t) * 1e3)
s2 = xl.make_slot(lane=1); s2.prepare(1024, 1024, 4, 0.6, use_controlnet=False)  # (explicit: this process already handed out lanes 0-3)
for e in (xl, s2):
    e.ops.upload(e.frame_u8, torch.from_numpy(f))
for i in range(4): (xl, s2)[i % 2].launch()
for e in (xl, s2): e.ops.synchronize()
t = time.perf_counter()
for i in range(12): (xl, s2)[i % 2].launch()
for e in (xl, s2): e.ops.synchronize()
fps2 = 12 / (time.perf_counter() - t)
row = {"config": "config4 SDXL 1024x1024 4-step (27.0 TFLOP/frame)", "p50_latency_ms_1_in_flight": round(sorted(lat)[4], 2),
       "fps_2_in_flight": round(fps2, 2), "mfma_frac_at_that_fps": round(27.04 * fps2 / 2500.0, 4)}
out.append(row)
print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(os.path.join("gpurun_out", "bench_configs.json"), "w"), indent=1)
