import time, torch, numpy as np, sys
sys.path.insert(0, '.')
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps
t=time.time()
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
print("weights", time.time()-t)
ops = HipOps(0)
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
text = (torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half()
eng.set_text_embeds(text)
import os
eng.overlap_controlnet = os.environ.get('OVL','1')=='1'
for cn in (False, True):
    t=time.time(); plan = eng.prepare(512, 512, 4, 0.6, use_controlnet=cn); print("prepare", time.time()-t, {k:v for k,v in plan.items() if k!='sizes'})
    f = np.random.default_rng(0).integers(0,256,(512,512,3),dtype=np.uint8)
    for _ in range(3): eng.infer_u8(f)
    ops.synchronize(); t=time.time()
    N=20
    for _ in range(N): eng.launch()
    ops.synchronize(); dt=(time.time()-t)/N
    print(f"cn={cn}: {dt*1e3:.2f} ms/frame  {1/dt:.1f} fps")
    slots=[eng]
    for S in (2,3,4):
        sl = eng.make_slot(); sl.prepare(512, 512, 4, 0.6, use_controlnet=cn); slots.append(sl)
        for e in slots: e.ops.upload(e.frame_u8, torch.from_numpy(f)); e.launch()
        for e in slots: e.ops.synchronize()
        t=time.time()
        for i in range(N*S): slots[i%S].launch()
        for e in slots: e.ops.synchronize()
        dt=(time.time()-t)/(N*S)
        print(f"   slots={S}: {dt*1e3:.2f} ms/frame  {1/dt:.1f} fps")
    if cn:
        for k, v in sorted(ops.tile_override.items()): print("   tuned", k, v)
    eng.prepare(512, 512, 4, 0.6, use_controlnet=cn, use_graph=False)
    ops.profile_begin(); eng.launch(); ops.synchronize(); st = ops.profile_end()
    for k,v in st.items(): print(f"   {k:14s} {v['ms']:8.3f} ms  {v['launches']:5d} launches  {v['flops']/1e12:.3f} TFLOP  -> {v['flops']/1e9/max(v['ms'],1e-9):.1f} TFLOP/s" )
