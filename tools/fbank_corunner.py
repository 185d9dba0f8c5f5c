#!/usr/bin/env python3
"""Which co-running kernel disturbs the mixture's Fbank? One captured graph: the main branch launches `delay` tiny kernels and then the
log-mel front-end of the mixture; a forked branch runs (a part of) the speaker branch's forward of the configs[0] model. Replayed N times;
a device-side counter (inside the graph) counts replays whose Fbank output is not bit-identical to the serial reference.
usage: python tools/fbank_corunner.py [replays] [side: speaker|frontend|fbank|none] [delays e.g. 0,2,4,8]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

entry = importlib.import_module("__graft_entry__")
ops = importlib.import_module("ts-asr_amd.ops")
from oracle.golden_recipe import golden_inputs  # noqa: E402
from test_model_gpu import make_batch  # noqa: E402

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
side_kind = sys.argv[2] if len(sys.argv) > 2 else "speaker"
delays = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,3,4,6,8,10,12,16").split(",")]
victim = sys.argv[4] if len(sys.argv) > 4 else "fbank"      # fbank | lds (LDS canary) | vgpr (register canary)
dev = "cuda"
brain, h = entry._config1_brain(dev, "bf16")
brain._setup_dtype()
brain.modules.train()
batch = make_batch(golden_inputs()).to(dev)
mixed, mixed_lens = batch.mixed_sig
enroll, enroll_lens = batch.enroll_sig
m = brain.modules


tokens_bos, tokens_bos_lens = batch.tokens_bos
_pred_in = {}


def side_work():
    with torch.no_grad():
        if side_kind == "none":
            return None
        if side_kind in ("predictor", "embedding", "lstm", "lstm_nomask", "proj", "gemm_in", "steps"):
            # the predictor of the captured step (recipes/tsasr.py::_predictor) and its pieces
            if side_kind in ("predictor", "embedding"):
                e = m.embedding(tokens_bos)
                if side_kind == "embedding":
                    return e
                d, _ = m.decoder(e, lengths=tokens_bos_lens)
                return m.decoder_proj(d)
            if "emb" not in _pred_in:
                _pred_in["emb"] = m.embedding(tokens_bos).clone()
                d0, _ = m.decoder(_pred_in["emb"], lengths=tokens_bos_lens)
                _pred_in["dec"] = d0.clone()
            if side_kind == "lstm":
                return m.decoder(_pred_in["emb"], lengths=tokens_bos_lens)[0]
            if side_kind == "lstm_nomask":
                return m.decoder(_pred_in["emb"])[0]
            if side_kind == "proj":
                return m.decoder_proj(_pred_in["dec"])
            rnn = m.decoder.rnn
            B, U, I = _pred_in["emb"].shape
            H = rnn.hidden_size
            if side_kind == "gemm_in":
                xp = torch.zeros(B * U, 32, dtype=torch.bfloat16, device=dev)
                wp = torch.zeros(4 * H, 32, dtype=torch.bfloat16, device=dev)
                return ops.gemm_bf16(xp, wp, B * U, 4 * H, 32, 32, 32, 0, 0, out_dtype=torch.float32)
            if side_kind == "steps":
                C = importlib.import_module("ts-asr_amd._capi")
                if "gates" not in _pred_in:
                    _pred_in["gates"] = torch.randn(B, U, H, 4, device=dev)
                    _pred_in["whh"] = torch.randn(4 * H, H, device=dev).to(torch.bfloat16)
                g = _pred_in["gates"].clone()
                c = torch.empty(B, U, H, device=dev)
                hh_ = torch.empty(B, U, H, dtype=torch.bfloat16, device=dev)
                ws = torch.empty(C.lib().tsasr_lstm_seq_workspace_bytes(B, U, H), dtype=torch.uint8, device=dev)
                C.check(C.lib().tsasr_lstm_seq_fwd(C.ptr(g), C.ptr(c), C.ptr(hh_), C.ptr(_pred_in["whh"]), B, U, H, C.BF16, C.ptr(ws), ws.numel(), C.stream_ptr()), "lstm")
                return hh_
        f = m.speaker_feature_extractor(enroll)
        if side_kind == "fbank":
            return f
        f = m.speaker_normalizer(f, enroll_lens, epoch=0)
        f = m.speaker_frontend(f)
        if side_kind == "frontend":
            return f
        return m.speaker_encoder(f, enroll_lens)


canary_err = torch.zeros(1, dtype=torch.int32, device="cuda")
canary_first = torch.zeros(4, dtype=torch.int32, device="cuda")


def main_work(delay):
    import ctypes
    Cc = importlib.import_module("ts-asr_amd._capi")
    with torch.no_grad():
        for _ in range(delay):
            ops.abs_lengths(mixed_lens, 100, 0)
        if victim == "lds":
            Cc.check(Cc.lib().tsasr_debug_lds_canary(512, 24576, 40, Cc.ptr(canary_err), Cc.ptr(canary_first), Cc.stream_ptr()), "canary")
        elif victim == "vgpr":
            Cc.check(Cc.lib().tsasr_debug_vgpr_canary(512, 60, Cc.ptr(canary_err), Cc.ptr(canary_first), Cc.stream_ptr()), "canary")
        return m.feature_extractor(mixed)


ref = main_work(0).clone()
side_work()
torch.cuda.synchronize()
side = torch.cuda.Stream()
for delay in delays:
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    worst = torch.zeros(1, dtype=torch.int32, device=dev)
    g = torch.cuda.CUDAGraph()
    main_work(delay)
    side_work()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            keep = side_work()
        out = main_work(delay)
        cur.wait_stream(side)
        n = (out != ref).sum().to(torch.int32).reshape(1)
        bad.add_((n > 0).to(torch.int32))
        torch.maximum(worst, n, out=worst)
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    print(f"side {side_kind:9s} delay {delay:2d}: {int(bad.item())} of {replays} replays with a wrong Fbank output (most differing values in one replay: {int(worst.item())})"
          + (f"; {victim} canary: {int(canary_err.item())} corrupted words, first {[hex(v & 0xffffffff) for v in canary_first.tolist()]}" if victim != "fbank" else ""), flush=True)
    if int(bad.item()) and victim == "fbank":      # what the last replay's output looks like when it is wrong: replay until one is
        for _ in range(400):
            g.replay()
            torch.cuda.synchronize()
            d = (out != ref).nonzero()
            if d.shape[0]:
                fr = sorted(set((int(a), int(b)) for a, b, _ in d.tolist()))
                print(f"    a wrong replay: {d.shape[0]} values in {len(fr)} frames (utt, frame): {fr[:24]}; bins of the first frame: "
                      f"{sorted(int(c) for a, b, c in d.tolist() if (int(a), int(b)) == fr[0])[:40]}; values "
                      + ", ".join(f"{float(out[tuple(k)]):.4f} vs {float(ref[tuple(k)]):.4f}" for k in d[:6].tolist()), flush=True)
                break
    canary_err.zero_()
    del g, keep, out
