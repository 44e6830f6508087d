"""Experiment: the dense two-launch step with the 256 episodes split into S groups on S streams (each group its own HIP graph
of 10 steps).  Each launch is latency-bound at one workgroup chain per CU; independent groups can overlap each other's chains."""
import sys, time
import torch
sys.path.insert(0, '.')
import bench

dev = torch.device('cuda:0')
B, N, K = 256, 100, 3
for S in (1, 2, 4, 8):
    groups = [bench.Rollout(dev, B // S, N, K, [32, 32], seed=1000 + g) for g in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    graphs = []
    for g, st in zip(groups, streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2):
                g.step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(10):
                g.step()
        graphs.append(gr)
    torch.cuda.synchronize()

    def run(n):
        for _ in range(n // 10):
            for gr, st in zip(graphs, streams):
                with torch.cuda.stream(st):
                    gr.replay()
    run(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 500
    run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("S=%d groups of %d episodes: %.2f us per step of all %d episodes -> %.3e agent-steps/s" % (S, B // S, 1e6 * el / steps, B, B * N * steps / el))
