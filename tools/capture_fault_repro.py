#!/usr/bin/env python
"""Repro attempt for the hipStreamEndCapture fault DESIGN.md section 8 reports (round 4): a plane loop whose side streams PERSIST over the
loop -- forked from the capturing stream once, ordered among themselves by events recorded inside the capture, joined once at the end --
captured into one HIP graph with torch operators only (no satmvs library in the process: /proc/self/maps is checked).

  side[0]  "encoder": runs ahead over the planes, records enc[d]
  side[1..L] "recurrent chain" of level l: waits for enc[d], carries its state from plane d-1 to d, records st[l][d]
  capturing stream "decoder": waits for every st[l][d], accumulates

Variants (argv[1]): persist (the shape described above), perplane (fork / join inside every iteration: what ships), nograph (eager);
persist_bwd / perplane_bwd / nostream_bwd (everything on the capturing stream): the weights require gradients and loss.backward() is captured too (autograd runs a node's backward on the
stream of its forward: the round-4 observation was made on the captured TRAINING step).
Prints the result checksum of 3 replays and "OK", or dies where the runtime does."""
import sys
import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "persist"
BWD = mode.endswith("_bwd")
mode = mode.replace("_bwd", "")
D = int(sys.argv[2]) if len(sys.argv) > 2 else 48
LEVELS = 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(D, 1, 16, 96, 192, device=dev)
w_enc = (torch.randn(16, 16, 3, 3, device=dev) * 0.1).requires_grad_(BWD)
w_rec = [(torch.randn(16, 32, 3, 3, device=dev) * 0.1).requires_grad_(BWD) for _ in range(LEVELS)]
conv = torch.nn.functional.conv2d


def loop(main):
    side = [main] * (1 + LEVELS) if mode == "nostream" else [torch.cuda.Stream(dev) for _ in range(1 + LEVELS)]
    states = [torch.zeros(1, 16, 96, 192, device=dev) for _ in range(LEVELS)]
    acc = torch.zeros(1, 16, 96, 192, device=dev)
    if mode != "perplane":
        for s in side:
            s.wait_stream(main)                                   # fork once
    for d in range(D):
        if mode == "perplane":
            for s in side:
                s.wait_stream(main)
        with torch.cuda.stream(side[0]):
            e = torch.relu(conv(x[d], w_enc, padding=1))
            enc = torch.cuda.Event(); enc.record(side[0])
        evs = []
        for l in range(LEVELS):
            with torch.cuda.stream(side[1 + l]):
                side[1 + l].wait_event(enc)
                states[l] = torch.tanh(conv(torch.cat((e, states[l]), 1), w_rec[l], padding=1))
                ev = torch.cuda.Event(); ev.record(side[1 + l]); evs.append(ev)
        for ev in evs:
            main.wait_event(ev)
        acc = acc + sum(states)
        if mode == "perplane":
            for s in side:
                main.wait_stream(s)
    if mode != "perplane":
        for s in side:
            main.wait_stream(s)                                   # join once
    if BWD:
        for w in [w_enc] + w_rec:
            w.grad = None
        acc.sum().backward()
        return torch.stack([w.grad.double().sum() for w in [w_enc] + w_rec]).sum() + acc.detach().double().sum()
    return acc


assert not any("satmvs" in l for l in open("/proc/self/maps")), "the library must not be mapped"
print("torch", torch.__version__, "hip", torch.version.hip, "mode", mode + ("_bwd" if BWD else ""), "planes", D, flush=True)
if mode == "nograph":
    out = loop(torch.cuda.current_stream())
    torch.cuda.synchronize()
    print("checksum %.6e" % float(out.double().sum()))
else:
    main = torch.cuda.Stream(dev)
    with torch.cuda.stream(main):
        loop(main)                                                # warm-up (MIOpen solver choice, allocator)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capturing ...", flush=True)
    with torch.cuda.graph(g, stream=main):
        out = loop(main)
    print("captured; replaying", flush=True)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        print("checksum %.6e" % float(out.double().sum()), flush=True)
print("OK")
