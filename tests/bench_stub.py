"""Stub engine for the CPU plumbing test of bench.py (tests/test_bench_cpu.py): every Engine method bench.py calls, on CPU tensors, with
outputs that are pure functions of the inputs -- so a clip's "generated ids" identify the pixels and the prompt that reached the LLM stage,
whatever path (pipelined / ragged batch / exchange over gloo / serial) carried them.  Run as a script it is one rank of a bench run:
  RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p GVL_BENCH_BACKEND=gloo python tests/bench_stub.py --gpus n [bench flags]
GVL_STUB_HANG_RANK=r makes rank r sleep inside its first vision encode of the timed region (a dead peer): the others must hit the watchdog."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa: E402,F401

HID, LSEG = 16, 5


class StubEngine:
    bench_hw = (8, 8)               # tiny frames: the pool of 2 x cps clips stays a few MB

    def __init__(self, dev):
        self.device = dev
        self.tokens_per_seg = LSEG
        self.seqs, self.next = {}, 0
        self.calls = 0
        self.hang_rank = int(os.environ.get("GVL_STUB_HANG_RANK", "-1"))
        self.rank = int(os.environ.get("RANK", "0"))
        if os.environ.get("GVL_STUB_COMM"):             # offer the communicator entry points (a real engine always has them)
            self.comm_unique_id, self.comm_init, self.comm_count, self.allgather_visual = self._comm_unique_id, self._comm_init, self._comm_count, self._allgather_visual
        if os.environ.get("GVL_STUB_COMM_HANG"):        # a libgvl communicator whose bootstrap never returns (bench's bounded end-of-run diagnostic)
            self.comm_unique_id = lambda: b"stub-unique-id"
            self.comm_init = lambda uid, rank, world: time.sleep(3600)

    # ---- vision: one row block per segment whose value encodes the segment's pixels ------------------------------------------------
    def _seg_code(self, sp, tp):
        return (sp.float().mean(dim=(1, 2, 3)) * 3.0 + tp.float().mean(dim=(1, 2, 3, 4)) * 5.0)            # [n]

    def _rows(self, code):
        n = code.shape[0]
        base = torch.arange(LSEG, dtype=torch.float32)[None, :, None] * 0.125
        return (code[:, None, None] + base).expand(n, LSEG, HID).reshape(n * LSEG, HID).to(torch.bfloat16)

    def _maybe_hang(self):
        self.calls += 1
        if self.rank == self.hang_rank and self.calls > int(os.environ.get("GVL_STUB_HANG_AFTER", "3")):
            time.sleep(3600)

    def encode_segments(self, sp, tp):
        self._maybe_hang()
        return self._rows(self._seg_code(sp, tp))

    def clip_encode(self, sp):
        self._maybe_hang()
        return sp.float().mean(dim=(1, 2, 3))[:, None, None]                                                # [n, 1, 1] "features"

    def iv2_encode(self, tp):
        return tp.float().mean(dim=(1, 2, 3, 4))[:, None, None]

    def build_visual(self, cf, vf):
        return self._rows(cf[:, 0, 0] * 3.0 + vf[:, 0, 0] * 5.0)

    # ---- LLM: the "answer" is a hash of the spliced prefix ------------------------------------------------------------------------
    def splice(self, ids, vis):
        k = ids.index(-200)
        row = lambda t: torch.full((1, HID), float(t % 251) / 16.0, dtype=torch.bfloat16)
        return torch.cat([row(t) for t in ids[:k]] + [vis] + [row(t) for t in ids[k + 1:]], 0)

    def seq_alloc(self, max_tokens):
        self.next += 1
        self.seqs[self.next] = None
        return self.next

    def seq_free(self, s):
        del self.seqs[s]

    def prefill(self, s, emb):
        w = torch.arange(1, emb.shape[0] + 1, dtype=torch.float64)
        self.seqs[s] = (int((emb[:, 0].double() * w).sum().item() * 16) % 100003, emb.shape[0])

    def prefill_batch(self, seqs, embs):
        for s, e in zip(seqs, embs):
            self.prefill(s, e)

    def decode_greedy(self, s, max_new, eos):
        h, n = self.seqs[s]
        return [(h * (i + 3) + n) % 32000 for i in range(max_new)]

    def decode_greedy_batch(self, seqs, max_new, eos):
        return [self.decode_greedy(s, max_new, eos) for s in seqs]

    # ---- "libgvl's own communicator": the C-ABI exchange of the real engine, played by gloo here (GVL_STUB_COMM=1) ---------------------
    def _comm_unique_id(self):
        return b"stub-unique-id".ljust(128, b"\0")

    def _comm_init(self, uid, rank, world):
        assert uid == self._comm_unique_id()
        self.comm_world = world

    def _comm_count(self):
        return self.comm_world

    def _allgather_visual(self, local):
        out = torch.empty((self.comm_world * local.shape[0], local.shape[1]), dtype=local.dtype)
        torch.distributed.all_gather_into_tensor(out, local.contiguous())
        self.gvl_gathers = getattr(self, "gvl_gathers", 0) + 1
        return out

    def kv_info(self):
        return {"total_pages": 0, "free_pages": 0, "pool_bytes": 0, "max_live_seqs": 0, "tokens": 0}


def factory(dev):
    return StubEngine(dev), types.SimpleNamespace(frames_per_seg=2, hidden=HID)


if __name__ == "__main__":
    import bench
    bench.main(sys.argv[1:], engine_factory=factory)
