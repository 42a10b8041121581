mkdir -p gpurun_out
bash tools/r04_profiles.sh a > gpurun_out/g14_profiles.log 2>&1
tail -60 gpurun_out/g14_profiles.log
