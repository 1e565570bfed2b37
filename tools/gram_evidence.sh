# Round-3 evidence of the Gram paths on the C4 table (2M x 64 fp64): kernel durations (rocprofv3 --kernel-trace), then separate --pmc
# passes (MFMA busy, FETCH_SIZE) for the plain, segmented and gathered forms.  Usage (GPU box): bash tools/gram_evidence.sh > gpurun_out/gram_paths.txt
bash tools/gram_paths.sh
bash tools/gram_orders.sh
for m in sse segments; do
  echo "== PMC, GRAM_MODE=$m"; GRAM_MODE=$m bash tools/gram_pmc.sh gram_pmc_$m | grep -v "^lds "
done
echo "== PMC, GRAM_MODE=gather (gram_gring_kernel only)"; GRAM_MODE=gather GRAM_KERNEL=gram_gring bash tools/gram_pmc.sh gram_pmc_gather | grep -v "^lds "
echo "== float table"; GRAM_DTYPE=f32 bash tools/gram_paths.sh
