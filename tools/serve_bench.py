"""Continuous batching at full size: N clips with RAGGED prompts and different answer lengths through
serve.ClipScheduler (ragged batched prefill, mixed-step batched decode, retirement between chunks), vision towers of the upcoming
clips on a second stream.   python tools/serve_bench.py [n_clips=24] [max_active=4] [chunk=8]
Prints clips/s and checks a sample of the answers against the one-at-a-time path (must be identical)."""
import os, sys, time, random
from collections import deque
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (engine construction + synthetic inputs of the benchmark)
from grounded_video_llm_amd import serve  # noqa: E402

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 24
max_active = int(sys.argv[2]) if len(sys.argv) > 2 else 4
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
eng, geo = bench.build_engine(dev, new_tokens=32, clips_per_step=max_active)
sp, tp, ids0 = bench.make_inputs(dev, 0)
rnd = random.Random(1)
reqs = []
for i in range(n_clips):                      # prompt length 60..140 tokens, answer budget 8..32 tokens
    n_text = rnd.randint(60, 140)
    gi = torch.Generator(); gi.manual_seed(100 + i)
    ids = torch.randint(3, 32000, (n_text,), generator=gi).tolist()
    ids[rnd.randint(5, n_text - 5)] = -200
    reqs.append((ids, rnd.randint(8, 32)))

sV, sL = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run():
    sch = serve.ClipScheduler(eng, None, max_active=max_active, chunk=chunk, max_prefill_rows=geo.max_prefill)
    ahead, out, nxt, rid2i = deque(), {}, 0, {}
    while nxt < n_clips or ahead or sch.pending():
        while nxt < n_clips and len(ahead) < max_active:          # vision of upcoming clips, asynchronously on its own stream
            with torch.cuda.stream(sV):
                vis = eng.encode_segments(sp, tp)
                ev = torch.cuda.Event(); ev.record(sV)
            ahead.append((nxt, vis, ev)); nxt += 1
        with torch.cuda.stream(sL):
            while ahead and (ahead[0][2].query() or not sch.pending()):
                i, vis, ev = ahead.popleft()
                sL.wait_event(ev); vis.record_stream(sL)
                rid = sch.submit(eng.splice(reqs[i][0], vis), reqs[i][1])
                rid2i[rid] = i
            sch.step()
            for rid, toks in list(sch.done.items()):
                out[rid2i[rid]] = toks
            sch.done.clear()
    torch.cuda.synchronize()
    return out, sch.stats


run()                                          # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
out, stats = run()
dt = time.perf_counter() - t0
tok = sum(len(v) for v in out.values())
print(f"serve: {n_clips} clips, ragged prompts (60-140 tok) / answers (8-32 tok), max_active={max_active} chunk={chunk}: "
      f"{n_clips / dt:.2f} clips/s, {tok / dt:.0f} answer tok/s, {dt * 1e3 / n_clips:.1f} ms/clip; stats {stats}")
# identical to the one-at-a-time path (same pixels for every clip here, so only the prompt / budget differ)
vis = eng.encode_segments(sp, tp)
for i in (0, n_clips // 2, n_clips - 1):
    ref = eng.generate_ids(eng.splice(reqs[i][0], vis), reqs[i][1], None)
    assert out[i] == ref, (i, out[i], ref)
print("serve: sampled answers identical to one-at-a-time generate")
