for cfg in "" "TOYNI_NT_MIN_BYTES=99999999999" "TOYNI_S3_PAIRS=1" "TOYNI_S3_PAIRS=1 TOYNI_NT_MIN_BYTES=99999999999"; do
  echo "== $cfg"
  env $cfg S3_LOGS=21 S3_BATCHES="16 128" timeout -k 10 200 python3 tools/s3check.py 2>&1 | grep -E "2\^21|OK|rror"
done
