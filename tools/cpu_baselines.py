# dev: the CPU baselines BASELINE.md section 3 asks for, timed on the GPU box's host cores
import os, sys, time
for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[k] = "1"
import numpy as np, multiprocessing as mp
sys.path.insert(0, '.')
N, FS = 32768, 1.25e6
def setup():
    from detprocess_amd import synth
    from oracle import of1x1 as orc
    pre = N // 2
    tmpl = synth.make_template(N, pre, FS); psd = synth.make_psd(N, FS)
    filt = orc.OFFilter(tmpl, psd, FS, pre)
    traces, _, _ = synth.make_traces(128, tmpl, psd, FS, filt.ampres, seed=os.getpid() % 1000)
    return orc, filt, traces
def worker(seconds):
    orc, filt, traces = setup()
    orc.process_events(filt, traces[:4], "unconstrained")
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.process_events(filt, traces, "unconstrained"); done += len(traces)
    return done / (time.perf_counter() - t0)
if __name__ == "__main__":
    ncores = len(os.sched_getaffinity(0))
    print("cores available:", ncores)
    print("1 core: %.0f traces/s" % worker(8.0))
    with mp.get_context("spawn").Pool(ncores) as pool:
        rates = pool.map(worker, [8.0] * ncores)
    print("%d cores (one process per core): %.0f traces/s total, %.0f per core" % (ncores, sum(rates), np.mean(rates)))
    # stronger baseline: batched scipy FFTs on all cores
    import scipy.fft as sf
    orc, filt, traces = setup()
    X = np.tile(traces, (8, 1))
    K = N // 2 + 1
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        V = sf.rfft(X, axis=1, workers=-1)
        A = sf.irfft(V * filt.Wf[:K], n=N, axis=1, workers=-1) * N
        chi0 = 2 * np.sum((V.real ** 2 + V.imag ** 2) * filt.g[:K], axis=1)
        idx = np.argmax(np.roll(A * A, filt.pre, axis=1), axis=1)
    dt = (time.perf_counter() - t0) / reps
    print("batched scipy.fft (workers=-1, no lowchi2): %.0f traces/s" % (X.shape[0] / dt))
