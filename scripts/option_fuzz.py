"""Random walks over the drop-in class's options in ONE process (programs evicted and rebuilt, lanes alternating): frame sizes
from 8 to 1024 a side incl. odd latent sizes and sizes that are not multiples of 8, 1-8 steps, strength 0.1-1.0, controlnet_scale
0-3, 1-3 frames per call, prompts as str / list, RGB / L / RGBA inputs.  Every call must return pictures of the right size, finite
and deterministic (the same call again: the same bits); option values the reference refuses must raise ValueError, nothing else.
usage (GPU box): python scripts/option_fuzz.py [seconds=120] [seed=0] [table]   (table: no timing runs for new shapes -- ten times
the calls per minute)"""
import os, sys, time
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.pipeline import VideoSDPipeline

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
p = VideoSDPipeline(model="SimianLuo/LCM_Dreamshaper_v7", controlnet="lllyasviel/control_v11p_sd15_canny", device=0,
                    **(dict(tuning_mode="table") if "table" in sys.argv[3:] else {}))
sides = [8, 16, 24, 40, 64, 72, 100, 128, 150, 192, 256, 360, 384, 432, 512, 640, 768, 1024]
t_end, n, refused = time.time() + seconds, 0, 0
while time.time() < t_end:
    w, h = int(rng.choice(sides)), int(rng.choice(sides))
    if w * h > 768 * 768:
        continue
    mode = str(rng.choice(["RGB", "RGB", "RGB", "L", "RGBA"]))
    sw, sh = int(rng.integers(8, 700)), int(rng.integers(8, 700))
    arr = rng.integers(0, 256, (sh, sw, {"RGB": 3, "L": 1, "RGBA": 4}[mode]), dtype=np.uint8)
    img = Image.fromarray(arr[..., 0] if mode == "L" else arr, mode)
    opts = dict(prompt=[str(rng.choice(["pixar, cg", "oil painting", "lego"]))] if rng.random() < 0.5 else str(rng.choice(["pixar, cg", "charcoal"])),
                height=h, width=w, strength=float(rng.choice([0.1, 0.3, 0.5, 0.6, 0.8, 1.0])), steps=int(rng.choice([1, 2, 3, 4, 8])),
                controlnet_scale=float(rng.choice([0.0, 0.5, 1.0, 3.0])), seed=int(rng.integers(0, 1000)))
    nb = int(rng.choice([1, 1, 2, 3]))
    desc = f"{w}x{h} {mode} from {sw}x{sh} steps {opts['steps']} strength {opts['strength']} cn {opts['controlnet_scale']} x{nb}"
    try:
        a = p.infer_batch([img] * nb, **opts)
        b = p.infer_batch([img] * nb, **opts)
    except ValueError as e:
        refused += 1
        print("refused:", desc, "--", str(e)[:80], flush=True)
        continue
    for x, y in zip(a, b):
        xa, ya = np.asarray(x), np.asarray(y)
        assert xa.shape == (h - h % 8, w - w % 8, 3), (desc, xa.shape)
        assert np.array_equal(xa, ya), "not deterministic: " + desc
    n += 1
    if n % 20 == 0:
        print(f"{n} calls ok, last: {desc}; programs cached {len(p._plans)}, evictions {p.evictions}", flush=True)
print(f"option fuzz passed: {n} calls, {refused} refused with ValueError")
