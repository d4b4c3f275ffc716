"""Probe: does multi-stream capture work on this torch/ROCm stack at all?  (run each case in its own process)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
case = sys.argv[1]
dev = torch.device("cuda")
if case == "torch2":
    a = torch.randn(512, 512, device=dev); b = torch.randn(512, 512, device=dev)
    s1 = torch.cuda.Stream()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            c = a @ b
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        s1.wait_stream(main)
        with torch.cuda.stream(s1):
            c = a @ b
        d = a + b
        main.wait_stream(s1)
        e = c + d
    g.replay(); torch.cuda.synchronize(); print("torch 2-stream capture OK", float(e.sum()))
elif case in ("A", "B", "C", "D"):
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.rdn_plan import rdn_forward, workspace
    from bin_amd import _lib as L
    from bin_amd.weights import reference_state_dict, synthetic_frames
    net = bin_stage4_lstm(); net.load_state_dict(reference_state_dict(0)); net = net.cuda().eval().set_precision("f16")
    fr = [f.cuda() for f in synthetic_frames(1, 1, 64, 64, 6)]
    m1 = net.model.model1_1
    kw = m1.kernel_weights(1)
    nb = L.lib().binhip_rdn_workspace_bytes(1, 64, 64, 2, 1, None)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ws1, ws2 = workspace(nb, dev, key="p1"), workspace(nb, dev, key="p2")
    cellm = net.clstm_4_prime

    def body():
        main = torch.cuda.current_stream()
        outs = []
        if case == "A":      # two RDN calls on two side streams, fork/join
            s1.wait_stream(main); s2.wait_stream(main)
            with torch.cuda.stream(s1):
                outs.append(rdn_forward(kw, [fr[0], fr[1]], ws=ws1))
            with torch.cuda.stream(s2):
                outs.append(rdn_forward(kw, [fr[1], fr[2]], ws=ws2))
            main.wait_stream(s1); main.wait_stream(s2)
        elif case == "B":    # one RDN call on one side stream
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                outs.append(rdn_forward(kw, [fr[0], fr[1]], ws=ws1))
            main.wait_stream(s1)
        elif case == "C":    # ConvLSTM cell on a side stream
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                outs.append(cellm(fr[0], None)[0])
            main.wait_stream(s1)
        elif case == "D":    # RDN on side stream 1, consumer RDN on side stream 2 (cross-stream dependency)
            s1.wait_stream(main); s2.wait_stream(main)
            with torch.cuda.stream(s1):
                a = rdn_forward(kw, [fr[0], fr[1]], ws=ws1)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                outs.append(rdn_forward(kw, [a, fr[2]], ws=ws2))
            main.wait_stream(s1); main.wait_stream(s2)
        return outs

    with torch.no_grad():
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o = body()
        g.replay(); torch.cuda.synchronize()
        print(f"case {case} capture OK", float(o[0].sum()))
else:
    from bin_amd.harness import GraphedNet
    from bin_amd.models.archs.RDN import bin_stage4_lstm
    from bin_amd.weights import reference_state_dict, synthetic_frames
    net = bin_stage4_lstm(); net.load_state_dict(reference_state_dict(0)); net = net.cuda().eval().set_precision("f16")
    frames = [f.cuda() for f in synthetic_frames(1, 1, 64, 64, 6)]
    net.n_streams = int(case)
    with torch.no_grad():
        g = GraphedNet(net, frames, multi_stream=True)
        out = g(*frames); torch.cuda.synchronize()
        ref = net(*frames)
        print(f"net capture with n_streams={case} OK, equal:", all(torch.equal(a, b) for a, b in zip(ref, out)))
