"""API-leg throughput against the client's queue depth (how many frames the caller keeps in flight)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

if __name__ == "__main__":
    frames = bench.synthetic_frames(12, 512, 512)
    for depth in [int(d) for d in (sys.argv[1:] or (6, 9, 12, 18))]:
        os.environ["VSD_API_DEPTH"] = str(depth)
        r = bench.api_leg(frames, n_frames=72)
        print(depth, r["api_fps"], r["api_p50_ms"], r["api_frames_per_launch"], r["api_stage_ms_p50"], flush=True)
