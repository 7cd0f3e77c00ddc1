"""How long does the CPU spend inside one hipGraphLaunch of the frame graph, and does launching the slots from separate
threads raise the frame rate?  (one frame per launch, 3 slots; then 3 frames per launch, 2 slots)"""
import os, sys, time, threading
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
for B, S in ((1, 3), (3, 2), (1, 1)):
    eng.overlap_controlnet = False
    eng.prepare(512, 512, 4, 0.6, use_controlnet=True, batch=B)
    slots = [eng]
    for _ in range(S - 1):
        sl = eng.make_slot(); sl.prepare(512, 512, 4, 0.6, use_controlnet=True, batch=B); slots.append(sl)
    for e in slots: e.launch()
    for e in slots: e.ops.synchronize()
    # (a) CPU time of one launch call
    cpu = []
    for i in range(6):
        e = slots[i % S]
        t = time.perf_counter(); e.launch(); cpu.append((time.perf_counter() - t) * 1e3)
    for e in slots: e.ops.synchronize()
    # (b) one thread launches everything
    n = 24
    t = time.perf_counter()
    for i in range(n): slots[i % S].launch()
    for e in slots: e.ops.synchronize()
    fps1 = n * B / (time.perf_counter() - t)
    # (c) one launcher thread per slot
    def worker(e, k):
        for _ in range(k): e.launch()
        e.ops.synchronize()
    ths = [threading.Thread(target=worker, args=(e, n // S)) for e in slots]
    t = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    fpsN = (n // S) * S * B / (time.perf_counter() - t)
    print(f"batch {B} slots {S}: hipGraphLaunch CPU time {np.median(cpu):.2f} ms (min {min(cpu):.2f}); fps one launcher thread {fps1:.1f}, one thread per slot {fpsN:.1f}", flush=True)
