"""In-node frame dispatch: one worker process per GPU, frames sharded round-robin, results released in
order, prompt embeddings broadcast over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in CPU tests).

Replaces the reference's Ray fan-out (/root/reference/diffusert/server.py:104-143, 273-277, 317-321):
  * `VideoSDPipeline.remote(**config)` -> `RemotePipeline`, whose `.infer.remote(img, **options)` is awaitable,
    so `img = await pipelines[gpu].infer.remote(frame.to_image(), **self.options)` (server.py:108) works as is;
  * `generating[gpu]` / first-idle scan -> `FrameDispatcher`: frame k goes to worker k mod N; a frame arriving
    while its worker is busy is dropped (server.py:132-137 drops while all are busy); completed frames are
    released in submission order ("in_order") or newest-wins ("latest", what server.py:117 effectively shows).
Frames never cross xGMI: each worker receives its frame from host memory and returns RGB to the host.
"""
import asyncio
import importlib
import multiprocessing as mp
import os
import threading
import time
from typing import Any, Callable, Dict, List, Optional

import torch


# ----------------------------------------------------------------------------------------- frame sharding
def shard_indices(n_frames: int, rank: int, world: int) -> List[int]:
    """Frame k is processed by rank k mod world (strict round-robin)."""
    return list(range(rank, n_frames, world))


def owner_of(frame_index: int, world: int) -> int:
    return frame_index % world


# ----------------------------------------------------------------------------------------- collective
def broadcast_prompt(embeds: Optional[torch.Tensor], header: Optional[Dict[str, float]] = None, src: int = 0,
                     device: Optional[torch.device] = None, shape=(77, 768)):
    """Rank `src` passes the prompt embeddings [77, cross_dim] (and an options header); every rank returns them.
    One small header broadcast + one 118 KB payload broadcast; the only collective on the path."""
    import torch.distributed as dist

    keys = ["epoch", "height", "width", "steps", "strength", "controlnet_scale", "seed"]
    rank = dist.get_rank()
    hdr = torch.zeros(len(keys), dtype=torch.float64, device=device)
    if rank == src:
        hdr.copy_(torch.tensor([float((header or {}).get(k, 0.0)) for k in keys], dtype=torch.float64))
        buf = embeds.to(device=device, dtype=torch.float16).reshape(shape).contiguous()
    else:
        buf = torch.zeros(shape, dtype=torch.float16, device=device)
    dist.broadcast(hdr, src=src)
    dist.broadcast(buf, src=src)
    return buf, {k: float(v) for k, v in zip(keys, hdr.tolist())}


# ----------------------------------------------------------------------------------------- worker process
def _resolve(path: str) -> Callable:
    mod, _, name = path.partition(":")
    return getattr(importlib.import_module(mod), name)


def _worker_main(conn, factory: str, config: Dict[str, Any], max_batch: int = 1):
    """Serve calls in order, like a Ray actor.  With max_batch > 1, `infer` calls that are ALREADY queued behind the one
    being taken (frames of other sessions, or of the same stream submitted ahead) and carry the same options are
    coalesced into one `infer_batch` launch: no waiting for a batch to fill, so a lone frame is never delayed.  When
    the pipeline has `submit_batch` / `collect_batch`, up to two launches are kept in flight (two engine lanes): while
    the GPU works on one, this process crops / resizes / uploads the next and converts / sends the previous one.
    Results always go back in request order."""
    try:
        pipe = _resolve(factory)(**config)
        conn.send(("ready", None))
    except BaseException as e:  # construction errors travel to the parent (the reference re-raises KeyError)
        conn.send(("error", (type(e).__name__, str(e))))
        return
    pipelined = max_batch > 1 and hasattr(pipe, "submit_batch") and hasattr(pipe, "collect_batch")
    backlog, inflight, lane = [], [], 0

    def fail(group, e):
        for r, _ in group:
            conn.send((r, False, (type(e).__name__, str(e))))

    def finish_oldest():
        group, handle = inflight.pop(0)
        try:
            for (r, _), o in zip(group, pipe.collect_batch(handle)):
                conn.send((r, True, o))
        except BaseException as e:
            fail(group, e)

    while True:
        if inflight and not backlog and not conn.poll(0):  # nothing new to start: hand back the oldest launch
            finish_oldest()
            continue
        msg = backlog.pop(0) if backlog else conn.recv()
        if msg is None:
            break
        rid, method, args, kwargs = msg
        group = [(rid, args)]
        batchable = max_batch > 1 and method == "infer" and len(args) == 1 and hasattr(pipe, "infer_batch")
        if batchable:
            while len(group) < max_batch and (backlog or conn.poll(0)):
                nxt = backlog.pop(0) if backlog else conn.recv()
                if nxt is not None and nxt[1] == "infer" and len(nxt[2]) == 1 and nxt[3] == kwargs:
                    group.append((nxt[0], nxt[2]))
                else:  # different options / another method / shutdown: serve it next, stop growing this batch
                    backlog.insert(0, nxt)
                    break
        if batchable and pipelined:
            try:
                handle = pipe.submit_batch([a[0] for _, a in group], lane=lane, **kwargs)
            except BaseException as e:
                while inflight:
                    finish_oldest()
                fail(group, e)
                continue
            lane ^= 1
            inflight.append((group, handle))
            if len(inflight) > 1:
                finish_oldest()
            continue
        while inflight:  # anything else runs alone, after what is in flight
            finish_oldest()
        try:
            if len(group) > 1:
                outs = pipe.infer_batch([a[0] for _, a in group], **kwargs)
                for (r, _), o in zip(group, outs):
                    conn.send((r, True, o))
            else:
                conn.send((rid, True, getattr(pipe, method)(*args, **kwargs)))
        except BaseException as e:
            fail(group, e)
    while inflight:
        finish_oldest()


class _RemoteMethod:
    def __init__(self, owner: "RemotePipeline", name: str):
        self._owner, self._name = owner, name

    def remote(self, *args, **kwargs) -> "asyncio.Future":
        """Awaitable (from inside a running event loop) or `.result()`-able concurrent future otherwise."""
        return self._owner._submit(self._name, args, kwargs)

    def __call__(self, *args, **kwargs):
        return self._owner._submit_sync(self._name, args, kwargs)


class RemotePipeline:
    """A VideoSDPipeline living in its own process (one per GPU), like the reference's Ray actor: calls are
    serialised per worker, inputs/outputs are pickled copies."""

    def __init__(self, factory: str = "videosd_amd.pipeline:VideoSDPipeline", start_timeout: float = 600.0, batch: int = 1,
                 **config):
        """batch > 1: the worker coalesces up to `batch` queued `infer` calls with equal options into one launch."""
        ctx = mp.get_context("spawn")
        self._conn, child = ctx.Pipe()
        self._proc = ctx.Process(target=_worker_main, args=(child, factory, config, int(batch)), daemon=True)
        self._proc.start()
        child.close()
        if not self._conn.poll(start_timeout):
            self.close()
            raise RuntimeError("pipeline worker did not start")
        tag, payload = self._conn.recv()
        if tag != "ready":
            self.close()
            exc = KeyError if payload[0] == "KeyError" else RuntimeError
            raise exc(f"pipeline worker failed to start: {payload[0]}: {payload[1]}")
        self._lock = threading.Lock()
        self._next = 0
        self._pending: Dict[int, Any] = {}
        # Requests leave through a writer thread: a frame is ~0.8 MB, far more than the pipe buffers, so a `send` in the
        # caller's thread would block the event loop while the worker is busy -- and while holding the lock the reader
        # needs to hand back the worker's (equally large) result, which is a deadlock.
        import queue

        self._outbox: "queue.Queue" = queue.Queue()
        self._writer = threading.Thread(target=self._write_loop, daemon=True)
        self._writer.start()
        self._reader = threading.Thread(target=self._read_loop, daemon=True)
        self._reader.start()
        self.infer = _RemoteMethod(self, "infer")
        self.compile_model = _RemoteMethod(self, "compile_model")
        self.set_prompt_embeds = _RemoteMethod(self, "set_prompt_embeds")

    def _write_loop(self):
        while True:
            msg = self._outbox.get()
            try:
                self._conn.send(msg)
            except (OSError, ValueError, BrokenPipeError):
                return
            if msg is None:
                return

    def _read_loop(self):
        while True:
            try:
                rid, ok, payload = self._conn.recv()
            except (EOFError, OSError):
                return
            with self._lock:
                fut = self._pending.pop(rid, None)
            if fut is None:
                continue
            target, loop = fut
            if ok:
                setter = lambda t=target, p=payload: (not t.done()) and t.set_result(p)  # noqa: E731
            else:
                setter = lambda t=target, p=payload: (not t.done()) and t.set_exception(RuntimeError(f"{p[0]}: {p[1]}"))  # noqa: E731
            if loop is not None:
                loop.call_soon_threadsafe(setter)
            else:
                setter()

    def _send(self, name, args, kwargs, target, loop):
        with self._lock:
            rid = self._next
            self._next += 1
            self._pending[rid] = (target, loop)
        self._outbox.put((rid, name, args, kwargs))

    def _submit(self, name, args, kwargs):
        try:
            loop = asyncio.get_running_loop()
            fut = loop.create_future()
        except RuntimeError:
            import concurrent.futures

            loop, fut = None, concurrent.futures.Future()
        self._send(name, args, kwargs, fut, loop)
        return fut

    def _submit_sync(self, name, args, kwargs):
        import concurrent.futures

        fut = concurrent.futures.Future()
        self._send(name, args, kwargs, fut, None)
        return fut.result()

    def close(self):
        try:
            if hasattr(self, "_outbox"):
                self._outbox.put(None)
                self._writer.join(timeout=2)
            else:
                self._conn.send(None)
        except Exception:
            pass
        if self._proc.is_alive():
            self._proc.join(timeout=5)
        if self._proc.is_alive():
            self._proc.terminate()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------- dispatcher
class FrameDispatcher:
    """Round-robin frame scheduler over N pipeline handles (RemotePipeline or anything with `.infer.remote`).

    submit(frame, **options) returns the frame's sequence number, or None when the owning worker is still busy
    (the frame is dropped, as the reference does while every GPU is generating).  Results come out of
    `await next_result()`: in submission order ("in_order") or whichever finished last ("latest")."""

    def __init__(self, pipelines: List[Any], mode: str = "in_order", depth: int = 1):
        """depth: frames one worker may hold at once (1 = the reference's `generating[gpu]` flag; set it to the
        workers' `batch` x launches in flight so that queued frames can be coalesced)."""
        assert mode in ("in_order", "latest")
        self.pipelines, self.mode = pipelines, mode
        self.n = len(pipelines)
        self.depth = max(1, int(depth))
        self.busy = [0] * self.n              # server.py:277 `generating`, as a count
        self.healthy = [True] * self.n
        self.seq = 0
        self.submitted = 0
        self.dropped = 0
        self._done: Dict[int, Any] = {}
        self._next_release = 0
        self._inflight = set()
        self._event = asyncio.Event()
        self.avg_gen_time = 0.4               # server.py:96 prior, updated as an EMA (server.py:113)

    def submit(self, frame, **options) -> Optional[int]:
        k = self.seq
        self.seq += 1
        gpu = owner_of(k, self.n)
        if not self.healthy[gpu]:
            gpu = next((g for g in range(self.n) if self.healthy[g] and self.busy[g] < self.depth), gpu)
        if self.busy[gpu] >= self.depth or not self.healthy[gpu]:
            self.dropped += 1
            return None
        self.busy[gpu] += 1
        ticket = self.submitted
        self.submitted += 1
        self._inflight.add(ticket)
        asyncio.ensure_future(self._run(ticket, gpu, frame, options))
        return ticket

    async def _run(self, ticket, gpu, frame, options):
        t0 = time.time()
        try:
            img = await self.pipelines[gpu].infer.remote(frame, **options)
            self._done[ticket] = img
        except Exception as e:  # a failed worker frees its slot (server.py:110-111) and is skipped afterwards
            self.healthy[gpu] = False
            self._done[ticket] = e
        finally:
            self.busy[gpu] -= 1
            self._inflight.discard(ticket)
        self.avg_gen_time = 0.95 * self.avg_gen_time + 0.05 * (time.time() - t0)
        self._event.set()

    async def next_result(self):
        """(ticket, image-or-exception)."""
        while True:
            if self.mode == "latest" and self._done:
                t = max(self._done)
                img = self._done.pop(t)
                for old in [k for k in self._done if k < t]:
                    self._done.pop(old)
                self._next_release = t + 1
                return t, img
            if self.mode == "in_order" and self._next_release in self._done:
                t = self._next_release
                self._next_release += 1
                return t, self._done.pop(t)
            self._event.clear()
            await self._event.wait()

    @property
    def pending(self) -> int:
        return len(self._inflight) + len(self._done)
