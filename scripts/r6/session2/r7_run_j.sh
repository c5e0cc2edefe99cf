O=gpurun_out/r7j; mkdir -p $O
python scripts/r6/diag_sh_stream.py > $O/diag_sh_stream_1M.json 2> $O/diag.err; tail -2 $O/diag.err; cat $O/diag_sh_stream_1M.json | tr -d '\n' | sed 's/},/},\n/g'; echo
python scripts/r6/diag_sh_stream.py 5000000 > $O/diag_sh_stream_5M.json 2> $O/diag5.err; tail -2 $O/diag5.err; cat $O/diag_sh_stream_5M.json | tr -d '\n' | sed 's/},/},\n/g'; echo
