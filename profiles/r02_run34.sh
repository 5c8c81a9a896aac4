#!/bin/bash
# Round-2 GPU call 34: final evidence of the round (same script as call 13, at the final kernel) - profiles of the headline and of the reference's scene, per-config table,
# bench lines (default form and the driver's form).
mkdir -p gpurun_out
timeout -k 10 500 bash profiles/run_profile.sh r02 > gpurun_out/r02_profile_h8.log 2>&1; tail -4 gpurun_out/r02_profile_h8.log
timeout -k 10 400 bash profiles/run_profile.sh r02_default14 --scene default14 --steps 600 > gpurun_out/r02_profile_d14.log 2>&1; tail -4 gpurun_out/r02_profile_d14.log
timeout -k 10 300 python profiles/bench_configs.py --big --out gpurun_out/r02_configs.json > gpurun_out/r02_configs.log 2>&1; cut -c1-200 gpurun_out/r02_configs.log
timeout -k 10 300 python bench.py > gpurun_out/r02_bench_n1.json 2>gpurun_out/r02_bench_n1.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r02_bench_n1.json
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_n1_driver_form.json 2>gpurun_out/r02_bench_n1_driver_form.err; cut -c1-300 gpurun_out/r02_bench_n1_driver_form.json
timeout -k 10 200 python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/r02_bench_cfg4_n1.json 2>gpurun_out/r02_bench_cfg4_n1.err; cut -c1-300 gpurun_out/r02_bench_cfg4_n1.json
timeout -k 10 300 python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/r02_bench_cfg5_n1.json 2>gpurun_out/r02_bench_cfg5_n1.err; cut -c1-300 gpurun_out/r02_bench_cfg5_n1.json
timeout -k 10 120 python - <<'PY' > gpurun_out/r02_host_copy_out.log 2>&1
import sys
sys.path.insert(0, "html5-canvas-raytracer_amd")
import rt_host
lib = rt_host.load_library()
for name in ("h8", "default14", "lcg64_ss1"):
    sc = rt_host.load_scene(name)
    ts = []
    for i in range(8):
        _, st = rt_host.render(3840, 2160, sc, lib=lib)
        ts.append((round(st.total_ms, 3), round(st.kernel_ms, 3)))
    print(name, "rt_render 3840x2160 (total_ms, kernel_ms) per call:", ts, rt_host.elapsed_report(st, lib))
PY
cat gpurun_out/r02_host_copy_out.log
timeout -k 10 200 python bench.py --scene default14 --no-cpu-baseline > gpurun_out/r02_bench_default14_n1.json 2>gpurun_out/r02_bench_default14_n1.err; cut -c1-300 gpurun_out/r02_bench_default14_n1.json
bash profiles/isa_histogram.sh 0 0 0 0 > gpurun_out/r02_isa_histogram_rt_trace_0000.txt; bash profiles/isa_histogram.sh 1 0 0 1 > gpurun_out/r02_isa_histogram_rt_trace_1001.txt; bash profiles/isa_histogram.sh 0 0 1 1 > gpurun_out/r02_isa_histogram_rt_trace_0011.txt
