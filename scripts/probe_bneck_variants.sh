# timing of the fused bottleneck under the diagnostic builds (scripts/build_variant.sh bn_X conv_bneck.hip -DBN_DIAG_X)
for v in ${VARIANTS:-"" WGS1 NOX NOSTORE NORES NOP2 NOMEM}; do
  if [ -z "$v" -o "$v" = shipped ]; then unset LVC_AMD_LIB; else export LVC_AMD_LIB=$PWD/build/variants/bn_$v.so; fi
  echo "== variant ${v:-shipped}"
  timeout 120 python scripts/probe_bneck.py time 2>&1 | grep "fused\|occupancy" | sort -u
done
unset LVC_AMD_LIB
