import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from PIL import Image
from videosd_amd.pipeline import VideoSDPipeline
p = VideoSDPipeline(model="SimianLuo/LCM_Dreamshaper_v7", controlnet="lllyasviel/control_v11p_sd15_canny", device=0)
rng = np.random.default_rng(0)
for (w, h) in ((1280, 720), (1920, 1080), (360, 640), (8, 8), (16, 2048)):
    img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
    t0 = time.time()
    a = np.asarray(p.infer(img, prompt="pixar, cg", height=h, width=w, strength=0.6, steps=2))
    t1 = time.time()
    b = np.asarray(p.infer(img, prompt="pixar, cg", height=h, width=w, strength=0.6, steps=2))
    t2 = time.time()
    two = p.infer_batch([img, img], prompt="pixar, cg", height=h, width=w, strength=0.6, steps=2)
    d = np.abs(np.asarray(two[0]).astype(int) - a.astype(int)).mean()
    print(f"{w}x{h}: out {a.shape} range {a.min()}..{a.max()} first {t1-t0:.2f}s replay {1e3*(t2-t1):.1f} ms deterministic {np.array_equal(a,b)} batch-of-2 frame 0 vs single: mean |diff| {d:.3f} LSB, frames equal {np.array_equal(np.asarray(two[0]), np.asarray(two[1]))}", flush=True)
