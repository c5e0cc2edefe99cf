O=gpurun_out/r7k; mkdir -p $O
python scripts/r6/diag_leg5.py > $O/leg5_alone.json 2>$O/e1; cat $O/leg5_alone.json; echo
python scripts/r6/diag_leg5.py --gc-off > $O/leg5_alone_gc_off.json 2>$O/e2; cat $O/leg5_alone_gc_off.json; echo
python scripts/r6/diag_leg5.py --headline-first > $O/leg5_after_headline.json 2>$O/e3; cat $O/leg5_after_headline.json; echo
