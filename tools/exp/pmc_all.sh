for spec in "100 4194368 n100" "160 2621504 n160" "512 2097216 n512" "256 2621504 n256" "1000 1048640 n1000" "2048 524352 n2048" "3000 262208 n3000" "4096 524352 n4096" "8192 131136 n8192"; do
  set -- $spec
  bash tools/exp/pmc_tile.sh $1 $2 $3 > gpurun_out/pmc_$3_r5.txt 2>&1
  grep -h "GB/s" gpurun_out/pmc_$3/*.log | head -2 >> gpurun_out/pmc_$3_r5.txt
  echo "== $3"; cat gpurun_out/pmc_$3_r5.txt
done
