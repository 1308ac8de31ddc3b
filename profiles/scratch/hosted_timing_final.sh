touch palettenerf_amd/csrc/frame.hip; PNR_EXTRA_HIPCC_FLAGS="-DPNR_HOSTED_TIMING" python -m palettenerf_amd.build 2>&1 | grep -E " error" | head -3
mkdir -p gpurun_out
{
echo "# profiles/hosted_timing.py on a -DPNR_HOSTED_TIMING build (round 3, end): time stamps inside the lookup launches of one lego frame."
echo "# 'ordinary workgroups' = the lookup itself (every 16th workgroup stamped), 'hosted' = the workgroups that finish the march's queued rays."
for pose in 5 10 16; do
echo "== pose step $pose"; HOSTED_TIMING_BRIEF=1 python profiles/hosted_timing.py --pose $pose $(seq 1 4 27) 2>&1 | grep iteration | cut -c1-260
done
echo "== pose step 5, iterations 5 and 15 in detail"
python profiles/hosted_timing.py --pose 5 5 15 2>&1 | grep -v "amdgpu.ids" | tail -14
} > gpurun_out/r03_hosted_timing.txt
cat gpurun_out/r03_hosted_timing.txt | head -40
