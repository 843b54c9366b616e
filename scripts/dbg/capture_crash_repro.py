"""Minimal torch-only reproduction attempts of the hipStreamEndCapture crash seen when one SIDE stream of a captured
step waits for an event recorded on ANOTHER side stream (engine.py: _flush join).  Each variant runs in its own process:
    python scripts/dbg/capture_crash_repro.py            -> runs all variants, prints rc per variant
    python scripts/dbg/capture_crash_repro.py VARIANT    -> runs one (may segfault)
"""
import subprocess
import sys

VARIANTS = ["tail_only", "mid_then_more", "mid_then_more_fresh_event", "cross_both_ways", "mid_no_more",
            "mid_then_more_via_origin", "cross_second_via_origin", "cross_first_via_helper", "cross_no_more_on_a",
            "cross_both_ways_eager_streams"]


def run(variant):
    import torch
    dev = torch.device("cuda", 0)
    x = torch.zeros(1 << 20, device=dev)
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(8)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        x.add_(1.0)                                   # k0 on the origin stream
        evs[0].record(main)
        A.wait_event(evs[0]); B.wait_event(evs[0])    # fork both side streams
        with torch.cuda.stream(A):
            x[:1000].mul_(2.0)                        # k1 on A
        if variant == "tail_only":                    # event at the END of A, waited for by B: the engine's tail join
            evs[1].record(A); B.wait_event(evs[1])
            with torch.cuda.stream(B):
                x[2000:3000].mul_(2.0)
        elif variant == "cross_second_via_origin":    # B waits for A's mid event; A's later wait for B goes through main
            evs[1].record(A); B.wait_event(evs[1])
            with torch.cuda.stream(B):
                x[2000:3000].mul_(2.0)
            with torch.cuda.stream(A):
                x[4000:5000].mul_(2.0)
            evs[2].record(B); main.wait_event(evs[2]); evs[3].record(main); A.wait_event(evs[3])
            with torch.cuda.stream(A):
                x[6000:7000].mul_(2.0)
        elif variant == "cross_first_via_helper":     # B's wait for A goes through a third forked stream
            Cs = torch.cuda.Stream()
            Cs.wait_event(evs[0])
            evs[1].record(A); Cs.wait_event(evs[1]); evs[4].record(Cs); B.wait_event(evs[4])
            with torch.cuda.stream(B):
                x[2000:3000].mul_(2.0)
            with torch.cuda.stream(A):
                x[4000:5000].mul_(2.0)
            evs[2].record(B); A.wait_event(evs[2])
            with torch.cuda.stream(A):
                x[6000:7000].mul_(2.0)
            evs[7].record(Cs); main.wait_event(evs[7])
        elif variant == "cross_no_more_on_a":         # zig-zag without work on A between the two waits
            evs[1].record(A); B.wait_event(evs[1])
            with torch.cuda.stream(B):
                x[2000:3000].mul_(2.0)
            evs[2].record(B); A.wait_event(evs[2])
            with torch.cuda.stream(A):
                x[6000:7000].mul_(2.0)
        elif variant in ("mid_then_more", "mid_then_more_fresh_event", "cross_both_ways", "mid_then_more_via_origin",
                         "cross_both_ways_eager_streams"):
            if variant == "mid_then_more_via_origin":
                evs[1].record(A); main.wait_event(evs[1]); evs[4].record(main); B.wait_event(evs[4])
            else:
                evs[1].record(A); B.wait_event(evs[1])    # event in the MIDDLE of A's captured work, waited for by B
            with torch.cuda.stream(B):
                x[2000:3000].mul_(2.0)
            with torch.cuda.stream(A):
                x[4000:5000].mul_(2.0)                # more work on A behind the event
            if variant in ("cross_both_ways", "cross_both_ways_eager_streams"):          # ... and A later waits for B (the engine's gradient tail)
                evs[2].record(B); A.wait_event(evs[2])
                with torch.cuda.stream(A):
                    x[6000:7000].mul_(2.0)
        elif variant == "mid_no_more":
            evs[1].record(A); B.wait_event(evs[1])
            with torch.cuda.stream(B):
                x[2000:3000].mul_(2.0)
                x[3000:4000].mul_(2.0)
        ea, eb = (torch.cuda.Event(), torch.cuda.Event()) if variant == "mid_then_more_fresh_event" else (evs[5], evs[6])
        ea.record(A); eb.record(B)
        main.wait_event(ea); main.wait_event(eb)      # join both
        x.add_(1.0)
    g.replay()
    torch.cuda.synchronize()
    print(variant, "ok", float(x[0]))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for v in VARIANTS:
            r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=120)
            last = (r.stdout.strip().splitlines() or [""])[-1]
            print(f"{v:32s} rc={r.returncode} {last}", flush=True)
