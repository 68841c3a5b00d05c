# round-4 profiles (profiles/round4/): every workload of the bench line on the final library - the flagship's traffic pair (K = 20 and K = 8),
# C4, C5 in fp32 and fp64, the polyline ensemble, long reaches (team kernel and, for comparison, the multi-pass kernel) - then the default bench line
mkdir -p gpurun_out/r4 gpurun_out/prof
bash tools/profile.sh c3_f64 --reaches 65536 --steps 20 --warmup 5 && echo "c3_f64 done"
bash tools/profile.sh c3_f64_k8 --reaches 65536 --steps 8 --warmup 5 && echo "c3_f64_k8 done"
bash tools/profile.sh c4_f64 --workload c4 --reaches 32768 --steps 16 --warmup 2 && echo "c4 done"
FS_F32=1 bash tools/profile.sh c5_f32 --workload c5 --dtype f32 --nodes 512 --reaches 131072 --steps 32 --warmup 4 && echo "c5_f32 done"
bash tools/profile.sh c5_f64 --workload c5 --dtype f64 --nodes 512 --reaches 131072 --steps 32 --warmup 4 && echo "c5_f64 done"
bash tools/profile.sh irr_f64 --workload irr --reaches 8192 --steps 16 --warmup 2 && echo "irr done"
bash tools/profile.sh long_f64 --workload long --nodes 16384 --reaches 8192 --steps 8 --warmup 2 && echo "long done"
FS_NO_TEAM=1 bash tools/profile.sh long_multipass_f64 --workload long --nodes 16384 --reaches 8192 --steps 8 --warmup 2 && echo "long multipass done"
python tools/refresh_profiles.py c3_f64 --second c3_f64_k8 > gpurun_out/r4/refresh_c3.txt
for t in c4_f64 c5_f32 c5_f64 irr_f64 long_f64 long_multipass_f64; do python tools/refresh_profiles.py $t > gpurun_out/r4/refresh_$t.txt; done
python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; echo "bench rc $?"; tail -c 400 gpurun_out/r4/bench_default.json
