mkdir -p gpurun_out/r2
for i in 1 2; do
MI355RT_LIB=tools/bin/libmi355rt_prev.so python tools/ab_flags.py prev=0 2>&1 | grep -v amdgpu.ids
python tools/ab_flags.py new=0 2>&1 | grep -v amdgpu.ids
done
for i in 1 2; do
AB_SCENE=reflection_test AB_SIZES=1920x1080 AB_REFL=4 MI355RT_LIB=tools/bin/libmi355rt_prev.so python tools/ab_flags.py prev=0 2>&1 | grep "start"
AB_SCENE=reflection_test AB_SIZES=1920x1080 AB_REFL=4 python tools/ab_flags.py new=0 2>&1 | grep "start"
AB_SCENE=clebsch AB_SIZES=3840x2160 MI355RT_LIB=tools/bin/libmi355rt_prev.so python tools/ab_flags.py prev=0 2>&1 | grep "start"
AB_SCENE=clebsch AB_SIZES=3840x2160 python tools/ab_flags.py new=0 2>&1 | grep "start"
done
