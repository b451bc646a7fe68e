#!/bin/bash
# turn gpurun_out/r02p (written by tools/gpu_r02_profiles.sh on the GPU box) into the committed summaries under profiles/
cd "$(dirname "$0")/.."
O=gpurun_out/r02p; R=profiles
k() { python3 tools/kstats.py "$O/$1" "$R/$2" "$3" > /dev/null; }
k kt_bench/b_kernel_stats.csv r02_rocprofv3_kernel_stats_bench.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --no-cpu-baseline (2^20 x 64 fp32_tc_cor; includes warm-up, accuracy and event-profile legs: 3 x 20 + 8 calls)"
k kt_policy1/p1_kernel_stats.csv r02_kstats_policy1_householder_tc_cor_after.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 4 --policy 1 (Householder engine, error-corrected bf16x3 block reflectors, collapsed tree)"
k kt_policy1n/p1n_kernel_stats.csv r02_kstats_policy1_householder_notc_after.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_notc 4 --policy 1 (Householder engine, exact fp32 MFMA block reflectors, collapsed tree)"
k kt_c3/c3_kernel_stats.csv r02_kstats_c3_tc_cor_2p20x128_one_panel.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 6 --n 128 (C3, auto policy: all 128 columns as one Cholesky-QR panel; two-block Cholesky in one launch)"
k kt_c3n/c3n_kernel_stats.csv r02_kstats_c3_notc_2p20x128_one_panel.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_notc 6 --n 128 (C3 notc, one panel)"
k kt_c3p/c3p_kernel_stats.csv r02_kstats_c3_tc_cor_2p20x128_panels_policy5.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 4 --n 128 --policy 5 (C3 through the 64-column panel path: bf16x3 coupling coefficients, plain stores for the updated panel)"
k kt_c3r/c3r_kernel_stats.csv r02_kstats_c3_tc_cor_2p20x128_one_panel_reorth.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 4 --n 128 --reorth"
k kt_c4/c4_kernel_stats.csv r02_kstats_2p23x64_tc_cor.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 4 --m 8388608 (C4's global shape on one GPU)"
k kt_notc/nc_kernel_stats.csv r02_kstats_2p20x64_notc.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_notc 6"
k kt_c5/c5_kernel_stats.csv r02_kstats_c5_cond1e8_reorth.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 4 --reorth --cond 1e8 (C5: latms cond 1e8 with the reference's singular-value draw, Reorthogonalize=true; matrix built on the CPU)"
k kt_c5n/c5n_kernel_stats.csv r02_kstats_c5_cond1e8_noreorth.csv "same matrix, Reorthogonalize=false"
k kt_reorth/ro_kernel_stats.csv r02_kstats_reorth_wellconditioned_2p20x64.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 4 --reorth (U(-1,1), both sweeps speculative, Q^T Q accumulated by the first apply)"
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $R/r02_pmc_hbm_traffic.json 2 > /dev/null
python3 tools/pmc_traffic.py $O/pmc_fetch_c3 $O/pmc_write_c3 $R/r02_pmc_hbm_traffic_c3_2p20x128.json 2 > /dev/null
python3 - <<'PY'
import json
p='profiles/r02_pmc_hbm_traffic_c3_2p20x128.json'; d=json.load(open(p))
d['command']=d['command'].replace('tools/prof_run.py fp32_tc_cor 3','tools/prof_run.py fp32_tc_cor 3 --n 128')
d['workload']='2^20 x 128 fp32_tc_cor, auto policy (one Cholesky-QR panel of 128 columns), per launch'
json.dump(d,open(p,'w'),indent=1)
PY
python3 tools/pmc_sq.py $R/r02_pmc_sq_counters.json $O/sq_a $O/sq_b > /dev/null
python3 tools/pmc_sq.py $R/r02_pmc_sq_counters_policy1_householder.json $O/sq_p1a $O/sq_p1b > /dev/null
cp $O/bench_default.json $R/r02_bench_default.json
cp $O/cpp_speed.csv $R/r02_cpp_speed_blockqr.csv
grep -v "socket.cpp\|amdgpu.ids" $O/dist_one_rank.txt > $R/r02_dist_one_rank_transport_cost.txt
cp $O/bench_workloads.txt $R/r02_bench_other_workloads.txt
grep "2^20" $O/wide_check.txt > $R/r02_c3_one_panel_vs_panels.txt
head -7 $R/r02_rocprofv3_kernel_stats_bench.csv; cat $R/r02_bench_other_workloads.txt; cat $R/r02_cpp_speed_blockqr.csv
python3 -c "
import json;d=json.loads(open('$R/r02_bench_default.json').read().strip().split('\n')[-1]);print('bench default:',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'])"
