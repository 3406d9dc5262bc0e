"""Direct LDS-tiled stencil against its own roofline (fp32 vector FMA, 157.3 Tflop/s) and against the FFT path:
    python tools/stencil_bench.py [--json out.json]
For each cubic PSF edge K: HIP-event time of mvsim_convolve_dev with method 2 (stencil) and method 1 (FFT passes) on a
device-resident volume -- 512^3 for K <= 15, a 256 x 256 x 64 sub-volume for the large PSFs (2 K^3 flop per voxel: a 63^3
PSF on 512^3 would be 67 Tflop per launch).  Useful flops = 2 K^3 N (the zero taps the kernel pads each PSF row with to a
multiple of 4 are not counted)."""
import argparse, importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mvs = importlib.import_module("multiview-simulation_amd")
synth = importlib.import_module("multiview-simulation_amd.synthetic")

FP32_PEAK_TFLOPS = 157.3   # MI355X vector fp32 (MI355X_MICROARCH.md)


def time_conv(ctx, d_in, d_out, dim, psf, method, reps):
    ctx.convolve_dev(d_in, dim, psf.copy(), d_out, method=method)
    ctx.synchronize()
    ctx.enable_timing(True)
    for _ in range(reps):
        ctx.convolve_dev(d_in, dim, psf.copy(), d_out, method=method)
    t = ctx.timings()
    ctx.enable_timing(False)
    return t["psf_ms"] + t["convolve_ms"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--edges", default="3,5,7,9,11,15,31,51,63")
    args = ap.parse_args()
    rows = []
    with mvs.Context(0) as ctx:
        big = synth.sphere_phantom(512)
        small = np.ascontiguousarray(synth.sphere_phantom(256)[96:160])
        for k in (int(e) for e in args.edges.split(",")):
            v = big if k <= 15 else small
            nz, ny, nx = v.shape
            dim = (nx, ny, nz)
            d_in = ctx.dev_alloc(v.nbytes); ctx.upload(d_in, v)
            d_out = ctx.dev_alloc(v.nbytes)
            psf = synth.hourglass_psf(k, sigma=(max(0.6, k / 25.0), max(0.6, k / 25.0), max(0.8, k / 6.3))) if k >= 15 \
                else synth.gaussian_psf(k, sigma=(k / 5.0, k / 5.0, k / 4.0))
            reps = 3 if k <= 31 else 2
            ms2 = time_conv(ctx, d_in, d_out, dim, psf, 2, reps)
            ms1 = time_conv(ctx, d_in, d_out, dim, psf, 1, reps)
            flops = 2.0 * k ** 3 * v.size
            tf = flops / (ms2 * 1e-3) / 1e12
            rows.append({"K": k, "volume": list(dim), "stencil_ms": ms2, "fft_ms": ms1, "flop": flops, "TFLOPs": tf,
                         "frac_fp32_peak": tf / FP32_PEAK_TFLOPS, "stencil_over_fft": ms2 / ms1})
            print(f"K={k:2d} {nx}x{ny}x{nz}: stencil {ms2:9.3f} ms = {tf:6.1f} Tflop/s ({tf / FP32_PEAK_TFLOPS:5.1%} of fp32 peak)   "
                  f"FFT {ms1:8.3f} ms   stencil/FFT {ms2 / ms1:7.2f}", flush=True)
            ctx.dev_free(d_in); ctx.dev_free(d_out)
    if args.json:
        json.dump({"peak_TFLOPs": FP32_PEAK_TFLOPS, "rows": rows}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
