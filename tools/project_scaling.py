"""PROJECTED (not measured) weak-scaling curve of the mirrored strategy on one 8 x MI355X node.

Only one GPU is reachable from the build environment, so this is a model, labelled as such wherever it is quoted.
Inputs that ARE measured: the single-GPU step time and the share of it that is backward (rocprofv3 breakdowns in
profiles/). Model of the rest:
  * gradients (bf16) are all-reduced bucket by bucket while backward runs; a ring / direct all-reduce of B bytes over
    N ranks takes  2 (N-1)/N * B / bus_bw + latency  with bus_bw bracketed by one xGMI link per direction (153 GB/s:
    a single ring) and by two concurrent rings (306 GB/s);
  * a bucket's all-reduce can start when its last gradient exists; buckets are produced at a uniform rate over the
    backward pass, so everything but the LAST bucket (the first layers, `first_bucket_bytes`) overlaps as long as the
    total collective time is shorter than backward;
  * while a collective kernel runs it occupies `comm_cus` of the 256 CUs, slowing the concurrent compute by that share.

    python tools/project_scaling.py            # prints the table kept in profiles/projected_scaling_v2.txt
"""
import argparse


def project(step_ms, bwd_share, grad_bytes, last_bucket_bytes, n, bus_bw, lat_us=25.0, comm_cus=32, n_buckets=4):
    if n == 1:
        return step_ms
    f = 2.0 * (n - 1) / n
    t_all = f * grad_bytes / bus_bw * 1e3 + n_buckets * lat_us * 1e-3          # ms, every bucket
    t_last = f * last_bucket_bytes / bus_bw * 1e3 + lat_us * 1e-3               # ms, exposed after backward ends
    bwd_ms = step_ms * bwd_share
    hidden = t_all - t_last
    overflow = max(0.0, hidden - bwd_ms)                                         # collectives longer than backward
    steal = min(hidden, bwd_ms) * comm_cus / 256.0                               # CUs taken from the overlapped compute
    return step_ms + t_last + overflow + steal


def main():
    ap = argparse.ArgumentParser()
    ap.parse_args()
    cfgs = [
        # name, measured 1-GPU ms/step, per-GPU batch, backward share, gradient bytes, last bucket bytes, buckets
        ("ResNet-50 bf16, batch 1024/GPU (bench.py default, 16 MB buckets)", 91.4, 1024, 0.66, 25.6e6 * 2, 8e6, 4),
        ("BERT-base MLM bf16, 64 x 512 tokens/GPU (16 MB buckets)", 32.3, 64, 0.64, 110e6 * 2, 8e6, 14),
    ]
    print("PROJECTED weak scaling on one 8 x MI355X node (model, NOT a measurement; see tools/project_scaling.py)")
    print("bus bandwidth bracket: 153 GB/s (one ring over single xGMI links) .. 306 GB/s (two concurrent rings)\n")
    for name, ms, batch, bshare, gbytes, last, nb in cfgs:
        print(name)
        print(f"  measured: 1 GPU {ms:.1f} ms/step = {batch / ms * 1e3:,.0f} samples/s")
        print("  N   step ms (153..306 GB/s)   samples/s whole job      efficiency")
        for n in (2, 4, 8):
            lo = project(ms, bshare, gbytes, last, n, 153e9, n_buckets=nb)
            hi = project(ms, bshare, gbytes, last, n, 306e9, n_buckets=nb)
            print(f"  {n}   {lo:6.2f} .. {hi:6.2f}          {n * batch / lo * 1e3:9,.0f} .. {n * batch / hi * 1e3:9,.0f}"
                  f"   {ms / lo * 100:5.1f} % .. {ms / hi * 100:5.1f} %")
        print()


if __name__ == "__main__":
    main()
