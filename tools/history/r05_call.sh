#!/bin/bash
# r05_call.sh <step>: ONE gpurun call of round 5 (every step under its own wall-clock guard; logs under gpurun_out/r05/<step>/).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r05_call.sh 1'
set -u
step=${1:?step}
out=gpurun_out/r05/$step
mkdir -p "$out"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # run <seconds> <log> <command...>: never let one command take the call with it
	local limit=$1 log=$2
	shift 2
	echo "== $* (limit ${limit}s)" | tee -a "$out/$log"
	timeout "$limit" "$@" >>"$out/$log" 2>&1
	echo "== rc $?" | tee -a "$out/$log"
}
case $step in
1)  # first contact of the round's new code: the whole GPU suite, then the two A/B timings
	run 420 pytest.txt python -m pytest tests -m gpu -x -q
	tail -5 "$out/pytest.txt"
	run 200 decode_exc.txt python tools/r05_time_decode_exc.py
	tail -30 "$out/decode_exc.txt"
	run 200 encode.txt python tools/r05_time_encode.py
	tail -4 "$out/encode.txt"
	;;
2)  # why is the decode with exceptions slow on narrow vectors: round 4's library against this one and its patch variants, same box
	for lib in r04 "" dec_table dec_plain dec_nostore dec_fence; do
		if [ -z "$lib" ]; then export -n ALPGPU_LIB; unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 120 decode_ab.txt python tools/r05_decode_ab.py
	done
	export ALPGPU_LIB=$PWD/build/variants/libalpgpu_dec_table.so
	run 200 pytest_table.txt python -m pytest tests/test_decode_gpu.py -x -q -k "patched or falp or synthetic"
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/decode_ab.txt"
	tail -3 "$out/pytest_table.txt"
	;;
3)  # the decode loops chosen per vector (no store waits for the store before it): round 4's library, this one, the slot-table patch variant
	for lib in r04 "" dec_table; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		WIDTHS=1,3,6,9,12,16,20,26,32,38,44,53 run 150 decode_ab.txt python tools/r05_decode_ab.py
	done
	export ALPGPU_LIB=$PWD/build/variants/libalpgpu_dec_table.so
	run 200 pytest_table.txt python -m pytest tests/test_decode_gpu.py tests/test_float_gpu.py -x -q --durations=6
	unset ALPGPU_LIB
	run 200 pytest_default.txt python -m pytest tests/test_decode_gpu.py tests/test_decode_sum_gpu.py tests/test_reference_gpu.py -x -q --durations=6
	for lib in r04 ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 100 bench_headline.txt python bench.py --no-extras --steps 20 --warmup 10
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/decode_ab.txt"
	tail -12 "$out/pytest_table.txt"; tail -12 "$out/pytest_default.txt"
	grep "^{" "$out/bench_headline.txt" | cut -c1-600
	;;
4)  # narrow vectors, two per workgroup: what slowed them (store waits as a throttle? the patch arm's code?) and the residency each build wants
	for lib in r04 "" dec_wait dec_nopatch; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		WIDTHS=1,2,3,4,6,8,10,12,16 run 150 decode_ab.txt python tools/r05_decode_ab.py
	done
	unset ALPGPU_LIB
	run 240 resid_default.txt python tools/r05_decode_resid.py
	export ALPGPU_LIB=$PWD/build/variants/libalpgpu_dec_table.so
	EXCS=20 run 150 resid_table.txt python tools/r05_decode_resid.py
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/decode_ab.txt"
	grep -v "^==\|amdgpu.ids" "$out/resid_default.txt"
	grep -v "^==\|amdgpu.ids" "$out/resid_table.txt"
	;;
5)  # the default build without a patch arm: residency by width for 1 / 2 / 4 vectors per workgroup, the decode tests, the float decode against round 4's
	run 300 resid.txt python tools/r05_decode_resid.py
	run 200 pytest.txt python -m pytest tests/test_decode_gpu.py tests/test_float_gpu.py tests/test_decode_sum_gpu.py -x -q
	for lib in r04 ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		for vpw in 1 2 4; do SWEEP_VPW=$vpw run 100 f32.txt python tools/time_decode_f32.py; done
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/resid.txt"
	tail -3 "$out/pytest.txt"
	grep -v "^==\|amdgpu.ids" "$out/f32.txt"
	;;
6)  # the 32-bit unpack / conversion of vectors of <= 32 bits against the build without it (same box), tests first
	run 200 pytest.txt python -m pytest tests/test_decode_gpu.py tests/test_reference_gpu.py tests/test_fuzz_gpu.py -x -q
	tail -3 "$out/pytest.txt"
	for lib in dec_nonarrow "" dec_nonarrow ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		WIDTHS=1,2,3,4,6,8,10,12,14,16,20,24,28,32,36 PADS=0,3,6 VPWS=1,2 run 200 resid.txt python tools/r05_decode_resid.py
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/resid.txt"
	;;
7)  # the pruned rowgroup search: parity first (every test that looks at rowgroup states), then search alone and encode, against the build without it
	run 400 pytest.txt python -m pytest tests/test_encode_gpu.py tests/test_reference_gpu.py tests/test_fuzz_gpu.py tests/test_async_init_gpu.py tests/test_dropin_gpu.py tests/test_recovery_gpu.py -x -q
	tail -4 "$out/pytest.txt"
	for lib in init_noprune "" init_noprune ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		for kind in mixed rd mixed_exc0 mixed_exc10; do run 60 init.txt python tools/time_init.py $kind 1048576; done
		run 200 encode.txt python tools/r05_time_encode.py
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/init.txt"
	grep -v "^==\|amdgpu.ids" "$out/encode.txt"
	;;
8)  # the pruned search (now a -D build) with its race fixed: parity, then what it buys where its incumbent is good (a column of one kind of rowgroup)
	export ALPGPU_LIB=$PWD/build/variants/libalpgpu_init_prune.so
	run 400 pytest.txt python -m pytest tests/test_encode_gpu.py tests/test_reference_gpu.py tests/test_fuzz_gpu.py tests/test_async_init_gpu.py tests/test_dropin_gpu.py tests/test_recovery_gpu.py -x -q
	tail -4 "$out/pytest.txt"
	for lib in "" init_prune "" init_prune; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		for kind in uniform2 mixed; do run 60 init.txt python tools/time_init.py $kind 1048576; done
		run 200 encode.txt python tools/r05_time_encode.py 1048576 uniform2
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/init.txt"
	grep -v "^==\|amdgpu.ids" "$out/encode.txt"
	;;
9)  # the whole GPU suite on the current build; the benchmark column under the residency pads; float encode ordered / unordered
	run 500 pytest.txt python -m pytest tests -m gpu -x -q
	tail -4 "$out/pytest.txt"
	run 200 pads.txt python tools/r05_headline_pads.py
	grep -v "^==\|amdgpu.ids" "$out/pads.txt"
	for u in 0 1 0 1; do ALPGPU_ENCODE_UNORDERED=$u run 120 f32enc.txt python tools/time_encode_f32.py; done
	grep -v "^==\|amdgpu.ids" "$out/f32enc.txt"
	;;
10) # float: the shortcut bound from a table (the sink ran a division four times per vector): SUM sink and decode, round 4's library beside
	run 200 pytest.txt python -m pytest tests/test_float_gpu.py tests/test_decode_sum_gpu.py -x -q
	tail -3 "$out/pytest.txt"
	for lib in r04 "" r04 ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 100 sink.txt python tools/time_sink_f32.py
		SWEEP_VPW=2 run 100 f32.txt python tools/time_decode_f32.py
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/sink.txt"
	grep -v "^==\|amdgpu.ids" "$out/f32.txt"
	;;
11) # the bench line as the driver runs it (does it fit the 8 KB tail? every key there?), then with a 2-rank shared-GPU run of the N > 1 path
	run 900 bench.txt python bench.py --steps 20 --warmup 5
	grep "^{" "$out/bench.txt" | tail -1 > "$out/bench.json"
	wc -c "$out/bench.json"
	cat "$out/bench.json"
	;;
12) # the round's profile of the library as built: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh), every command under its own guard there
	timeout 1500 bash tools/profile_round.sh r05 > "$out/profile_round.txt" 2>&1
	echo "rc $?"; tail -5 "$out/profile_round.txt"
	;;
13) run 200 pairs.txt python tools/r05_pairs_on_mixed.py
	grep -v "^==\|amdgpu.ids" "$out/pairs.txt"
	;;
14) # hopeless rounds of the (e,f) search end at their checkpoint: parity, then the search alone and the encodes against the build without it
	run 400 pytest.txt python -m pytest tests/test_encode_gpu.py tests/test_float_gpu.py tests/test_reference_gpu.py tests/test_fuzz_gpu.py tests/test_async_init_gpu.py tests/test_dropin_gpu.py tests/test_recovery_gpu.py tests/test_sharding_gpu.py -x -q
	tail -4 "$out/pytest.txt"
	for lib in init_noearly "" init_noearly ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		for kind in mixed rd; do run 60 init.txt python tools/time_init.py $kind 1048576; run 60 init.txt python tools/time_init.py $kind 1048576 f32; done
		run 200 encode.txt python tools/r05_time_encode.py
		run 120 f32enc.txt python tools/time_encode_f32.py
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/init.txt"
	grep -v "^==\|amdgpu.ids" "$out/encode.txt" | cut -c1-330
	grep -v "^==\|amdgpu.ids" "$out/f32enc.txt"
	;;
15) # the unordered mode with one reservation per VECTOR (no tile barrier): parity, then timings
	run 400 pytest.txt python -m pytest tests/test_encode_gpu.py tests/test_float_gpu.py tests/test_recovery_gpu.py tests/test_async_init_gpu.py -x -q
	tail -4 "$out/pytest.txt"
	run 200 encode.txt python tools/r05_time_encode.py
	for u in 0 1; do ALPGPU_ENCODE_UNORDERED=$u run 120 f32enc.txt python tools/time_encode_f32.py; done
	grep -v "^==\|amdgpu.ids" "$out/encode.txt" | cut -c1-330
	grep -v "^==\|amdgpu.ids" "$out/f32enc.txt"
	;;
16) run 300 pytest.txt python -m pytest tests/test_encode_gpu.py tests/test_recovery_gpu.py tests/test_async_init_gpu.py -x -q
	tail -3 "$out/pytest.txt"
	run 200 encode.txt python tools/r05_time_encode.py
	grep -v "^==\|amdgpu.ids" "$out/encode.txt" | cut -c1-330
	;;
17) # per-stage instruction budget of k_sink_direct_f32: stop-at-stage builds under the SQ counters, three columns
	for kind in clean mixed rd; do
		for lib in sinkf_stop1 sinkf_stop2 sinkf_stop3 sinkf_stop4 sinkf_stop5 sinkf_stop6 ""; do
			if [ -z "$lib" ]; then unset ALPGPU_LIB; tag=full; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; tag=$lib; fi
			timeout 120 bash tools/pmc_busy.sh r05_${kind}_$tag python tools/r05_sinkf_counters.py $kind > /dev/null 2>&1
			echo "$kind $tag: $(grep k_sink_direct_f32 gpurun_out/pmcb_r05_${kind}_$tag.txt | head -1)" | tee -a "$out/sinkf_budget.txt"
		done
	done
	unset ALPGPU_LIB
	;;
18) # float exception values: LDS read + rare HBM read instead of one flat load per exception: tests, sink and decode against round 4's library (alternating)
	run 300 pytest.txt python -m pytest tests/test_float_gpu.py tests/test_decode_sum_gpu.py tests/test_fuzz_gpu.py tests/test_reference_gpu.py -x -q
	tail -3 "$out/pytest.txt"
	for lib in r04 "" r04 ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 100 sink.txt python tools/time_sink_f32.py
		SWEEP_VPW=2 run 100 f32.txt python tools/time_decode_f32.py
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/sink.txt" | cut -c1-120
	grep -v "^==\|amdgpu.ids" "$out/f32.txt"
	;;
19) for lib in f32head "" f32head ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 100 sink.txt python tools/time_sink_f32.py
		SWEEP_VPW=2 run 100 f32.txt python tools/time_decode_f32.py
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/sink.txt" | cut -c1-120
	grep -v "^==\|amdgpu.ids" "$out/f32.txt"
	;;
20) for kind in clean mixed rd; do run 60 cols.txt python tools/r05_sinkf_counters.py $kind; done
	grep -v "^==\|amdgpu.ids" "$out/cols.txt"
	;;
21) # float: exception values without a branch per value: parity, then sink / decode against the previous form (alternating)
	run 300 pytest.txt python -m pytest tests/test_float_gpu.py tests/test_decode_sum_gpu.py tests/test_fuzz_gpu.py tests/test_reference_gpu.py tests/test_last_register_gpu.py -x -q
	tail -3 "$out/pytest.txt"
	for lib in f32head "" f32head ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 100 sink.txt python tools/time_sink_f32.py
		for vpw in 2 4; do SWEEP_VPW=$vpw run 100 f32.txt python tools/time_decode_f32.py; done
	done
	unset ALPGPU_LIB
	grep -v "^==\|amdgpu.ids" "$out/sink.txt" | cut -c1-120
	grep -v "^==\|amdgpu.ids" "$out/f32.txt"
	;;
22) # Infinity Cache: decode with its inputs resident / flushed / pre-read; segmented decode of a 1 Mi-vector column with the next segments pre-read
	PARTS=1 run 400 mall.txt python tools/r05_mall_warm.py
	PARTS=2 run 500 mall.txt python tools/r05_mall_warm.py
	grep -v "^==\|amdgpu.ids" "$out/mall.txt"
	;;
23) # the read-ahead kernel beside the decode: parity, then shapes x pads x windows per width; two more grids on a few widths
	run 300 pytest.txt python -m pytest tests/test_decode_gpu.py -x -q -k "read_ahead or tuning or synthetic"
	tail -3 "$out/pytest.txt"
	run 600 ra.txt python tools/r05_read_ahead.py
	for g in 32 256; do
		ALPGPU_READ_AHEAD_GRID=$g WIDTHS=4,16,36,mix EXCS=0 WINDOWS=96 run 200 ra.txt python tools/r05_read_ahead.py
	done
	grep -v "^==\|amdgpu.ids" "$out/ra.txt"
	;;
24) # read-ahead, one poll per workgroup and ~7 us: what the pacing alone costs (mode 1), what unpaced reads do (mode 2), grids
	run 200 pytest.txt python -m pytest tests/test_decode_gpu.py -x -q -k "read_ahead"
	tail -3 "$out/pytest.txt"
	for g in 128 32; do for m in 0 1 2; do
		echo "== grid $g mode $m" >>"$out/ra.txt"
		ALPGPU_READ_AHEAD_GRID=$g ALPGPU_READ_AHEAD_MODE=$m WIDTHS=2,4,16,36,mix EXCS=0 PADS=0,14 WINDOWS=96,32 run 200 ra.txt python tools/r05_read_ahead.py
	done; done
	grep -v "amdgpu.ids\|== rc\|== python" "$out/ra.txt"
	;;
25) # what a nearly idle kernel on the second stream costs the decode: pace-only (mode 1) and sleep-only (mode 4) by grid and stream priority
	for cfg in "1 8 high" "1 32 high" "1 128 high" "4 8 high" "4 32 high" "4 128 high" "4 128 normal" "4 128 low" "1 128 low" "0 32 low"; do
		set -- $cfg
		ALPGPU_READ_AHEAD_MODE=$1 ALPGPU_READ_AHEAD_GRID=$2 ALPGPU_INIT_STREAM_PRIO=$3 run 100 beside.txt python tools/r05_beside.py
	done
	grep "^mode" "$out/beside.txt"
	;;
26) # (history: mode bit 3 and ALPGPU_READ_AHEAD_NO_REPORT were removed from the library after this call — both arms only made the read-ahead wait out its patience)
    # the pacing's cost taken apart (grid 128): polls of a word nobody writes; no reports from the decode; naps of 27 / 110 us; reports but no polls
	for cfg in "1 0" "9 0" "1 1" "2049 0" "8193 0" "9 1"; do
		set -- $cfg
		if [ "$2" = 1 ]; then export ALPGPU_READ_AHEAD_NO_REPORT=1; else unset ALPGPU_READ_AHEAD_NO_REPORT; fi
		echo "== mode $1 no_report $2" >>"$out/beside.txt"
		ALPGPU_READ_AHEAD_MODE=$1 ALPGPU_READ_AHEAD_GRID=128 run 100 beside.txt python tools/r05_beside.py
	done
	grep "^mode\|^== mode" "$out/beside.txt"
	;;
27) # read-ahead by fat workgroups (8 / 16 wavefronts), naps of 14 / 27 us between polls: net effect per width, with the pace-only reference
	for cfg in "1024 32 -" "2048 32 -" "1025 32 -" "1024 16 -" "1024 16 ra16" "2048 8 ra16"; do
		set -- $cfg
		if [ "$3" = - ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$3.so; fi
		echo "== mode $1 grid $2 lib $3" >>"$out/ra.txt"
		ALPGPU_READ_AHEAD_MODE=$1 ALPGPU_READ_AHEAD_GRID=$2 WIDTHS=2,4,16,36,mix EXCS=0,20 PADS=0,14 WINDOWS=96 run 200 ra.txt python tools/r05_read_ahead.py
	done
	unset ALPGPU_LIB
	grep -v "amdgpu.ids\|== rc\|== python" "$out/ra.txt"
	;;
28) # read-ahead on narrow vectors, every width up to 12 bits: grids 32 / 64 of eight wavefronts, naps of 14 us
	for g in 32 64; do
		echo "== grid $g" >>"$out/ra.txt"
		ALPGPU_READ_AHEAD_MODE=1024 ALPGPU_READ_AHEAD_GRID=$g WIDTHS=1,2,3,4,5,6,8,10,12 EXCS=0,20 PADS=0,3 WINDOWS=96,32 run 400 ra.txt python tools/r05_read_ahead.py
	done
	grep -v "amdgpu.ids\|== rc\|== python" "$out/ra.txt"
	;;
29) # the read-ahead's window per width and shape (grid 64 and 32 of eight wavefronts)
	for g in 64 32; do
		ALPGPU_READ_AHEAD_MODE=1024 ALPGPU_READ_AHEAD_GRID=$g run 500 win.txt python tools/r05_read_ahead_windows.py
	done
	grep -v "amdgpu.ids\|== rc\|== python" "$out/win.txt"
	;;
30) # the read-ahead paced by the clock: parity, then the lead (microseconds) per width and shape, grids 64 / 32 / 16 of eight wavefronts
	run 200 pytest.txt python -m pytest tests/test_decode_gpu.py -x -q -k "read_ahead"
	tail -3 "$out/pytest.txt"
	for g in 64 32 16; do
		ALPGPU_READ_AHEAD_GRID=$g run 500 win.txt python tools/r05_read_ahead_windows.py
	done
	grep -v "amdgpu.ids\|== rc\|== python" "$out/win.txt"
	;;
31) # narrow columns under the read-ahead: residency pad x lead
	for g in 64 128; do
		ALPGPU_READ_AHEAD_GRID=$g run 500 pads.txt python tools/r05_read_ahead_pads.py
	done
	grep -v "amdgpu.ids\|== rc\|== python" "$out/pads.txt"
	;;
32) # the read-ahead chosen by the library: decode tests, where it starts to pay by column length, the bench line
	run 400 pytest.txt python -m pytest tests/test_decode_gpu.py tests/test_sharding_gpu.py tests/test_dropin_gpu.py -x -q
	tail -3 "$out/pytest.txt"
	run 300 short.txt python tools/r05_read_ahead_short.py
	grep -v "amdgpu.ids\|^==" "$out/short.txt"
	run 400 bench.json python bench.py --steps 20 --warmup 5
	tail -c 6000 "$out/bench.json"
	;;
33) # by column length, cold (2 GiB read between launches)
	COLD=1 run 400 short.txt python tools/r05_read_ahead_short.py
	grep -v "amdgpu.ids\|^==" "$out/short.txt"
	;;
34) # columns that are not narrow: descriptors of all vectors + records of the narrow ones only, by grid
	for g in 16 32 64; do for b in 0 7 12 128; do
		ALPGPU_READ_AHEAD_GRID=$g ALPGPU_READ_AHEAD_BITS=$b run 100 mix.txt python tools/r05_read_ahead_mix.py
	done; done
	grep "^grid" "$out/mix.txt"
	;;
35) run 300 stride.txt python tools/r05_stride.py
	grep -v "amdgpu.ids\|^==" "$out/stride.txt"
	;;
36) # look-back that does not read an arrived word again: default library against the variant, alternating
	for lib in "" repoll "" repoll; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 120 enc.txt python tools/r05_time_encode.py
	done
	unset ALPGPU_LIB
	grep "^lib\|^mixed\|^rd" "$out/enc.txt" | cut -c1-330
	;;
37) # decode by segments: the decode tests, the bimodal column and friends through the plan
	run 400 pytest.txt python -m pytest tests/test_decode_gpu.py tests/test_container_gpu.py tests/test_host_pipeline_gpu.py -x -q
	tail -5 "$out/pytest.txt"
	run 300 seg.txt python tools/r05_segments.py
	grep -v "amdgpu.ids\|^==" "$out/seg.txt"
	;;
38) run 300 seg.txt python tools/r05_segments.py
	grep -v "amdgpu.ids\|^==" "$out/seg.txt"
	;;
39) # the whole GPU suite three times in a row (flakiness of the round's new tests), then smoke
	for i in 1 2 3; do run 600 pytest.txt python -m pytest tests -m gpu -q -x; tail -2 "$out/pytest.txt"; done
	run 120 smoke.txt python __graft_entry__.py smoke
	tail -1 "$out/smoke.txt"
	;;
40) # the plan's key (a caching allocator reuses buffers): decode tests, then the five columns again
	run 300 pytest.txt python -m pytest tests/test_decode_gpu.py tests/test_container_gpu.py -x -q
	tail -2 "$out/pytest.txt"
	run 300 seg.txt python tools/r05_segments.py
	grep -v "amdgpu.ids\|^==" "$out/seg.txt"
	;;
41) # float store decode without the store waits (a staged-exceptions instance of the quad): parity, then A/B against the build before, alternating
	run 300 pytest.txt python -m pytest tests/test_float_gpu.py tests/test_decode_sum_gpu.py -x -q
	tail -2 "$out/pytest.txt"
	for lib in f32old "" f32old ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 120 f32.txt python tools/r05_time_float_decode.py
	done
	unset ALPGPU_LIB
	grep -v "amdgpu.ids\|^==" "$out/f32.txt"
	;;
42) # float: 256 exception values staged per vector instead of 128: parity of the variant, then A/B alternating (decode and SUM)
	ALPGPU_LIB=$PWD/build/variants/libalpgpu_f32stage256.so run 300 pytest.txt python -m pytest tests/test_float_gpu.py tests/test_decode_sum_gpu.py -x -q
	tail -2 "$out/pytest.txt"
	for lib in "" f32stage256 "" f32stage256; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 120 f32.txt python tools/r05_time_float_decode.py
	done
	unset ALPGPU_LIB
	grep -v "amdgpu.ids\|^==" "$out/f32.txt"
	;;
43) # the lean encode's stores from registers of their own (no store waits for the one before it): parity, then A/B against the one-by-one form, alternating
	run 400 pytest.txt python -m pytest tests/test_encode_gpu.py tests/test_recovery_gpu.py tests/test_async_init_gpu.py tests/test_reference_gpu.py -x -q
	tail -2 "$out/pytest.txt"
	for lib in onebyone "" onebyone ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 120 enc.txt python tools/r05_time_encode.py
	done
	unset ALPGPU_LIB
	grep "^lib\|^mixed\|^rd" "$out/enc.txt" | cut -c1-330
	;;
44) run 200 sum.txt python tools/r05_sum_exc.py
	grep -v "amdgpu.ids\|^==" "$out/sum.txt"
	;;
45) # double SUM sink: the exception lookup pinned behind the pair's values (fewer spills): parity, then A/B alternating
	run 300 pytest.txt python -m pytest tests/test_decode_sum_gpu.py tests/test_last_register_gpu.py -x -q
	tail -2 "$out/pytest.txt"
	for lib in nopin "" nopin ""; do
		if [ -z "$lib" ]; then unset ALPGPU_LIB; else export ALPGPU_LIB=$PWD/build/variants/libalpgpu_$lib.so; fi
		run 120 sum.txt python tools/r05_sum_exc.py
	done
	unset ALPGPU_LIB
	grep -v "amdgpu.ids\|^==" "$out/sum.txt"
	;;
46) run 300 pytest.txt python -m pytest tests/test_decode_gpu.py -x -q
	tail -2 "$out/pytest.txt"
	;;
47) # the read-ahead's lead per width, finer, twice
	for i in 1 2; do
		WIDTHS=1,2,3,4,5,6,7 EXCS=0,20 WINDOWS=15,20,30,40,50,60,80 run 400 win.txt python tools/r05_read_ahead_windows.py
	done
	grep -v "amdgpu.ids\|^==" "$out/win.txt"
	;;
48) # the lead by width (option 13 = 0, the default) against 40 us fixed, decode tests first
	run 300 pytest.txt python -m pytest tests/test_decode_gpu.py -x -q
	tail -2 "$out/pytest.txt"
	for i in 1 2; do
		WIDTHS=1,2,3,4,5,6,7 EXCS=0,20 WINDOWS=0,40 run 300 win.txt python tools/r05_read_ahead_windows.py
	done
	grep -v "amdgpu.ids\|^==" "$out/win.txt"
	;;
49) # beyond 7 bits: the read-ahead (lead by width, and fixed leads) on 8-12-bit vectors, with and without exceptions, twice
	for i in 1 2; do
		WIDTHS=7,8,9,10,11,12 EXCS=0,20 WINDOWS=0,50,70,90 run 400 win.txt python tools/r05_read_ahead_windows.py
	done
	grep -v "amdgpu.ids\|^==" "$out/win.txt"
	;;
50) # the rule extended to 11 bits (9 with exceptions), leads 12 + 6.5 us per bit up to 60: decode tests, then the bench line
	run 300 pytest.txt python -m pytest tests/test_decode_gpu.py -x -q
	tail -2 "$out/pytest.txt"
	run 400 bench.json python bench.py --steps 20 --warmup 5
	python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05/50/bench.json") if l.startswith("{")][-1])
x = d["extras"]
print(d["value"], d["roofline"]["frac"])
for k in x:
    if "sweep" in k:
        print(k, x[k]["summary"])
        print("  on ", x[k]["frac"][:12])
        print("  off", x[k]["no_read_ahead"])
print(x["decode_bimodal"])
PY
	;;
51) # fewer search workgroups than CUs beside the encode (ALP columns): 64 / 96 / 128 / 192 against 256
	export ALPGPU_LIB=$PWD/build/variants/libalpgpu_initbase.so
	for b in 256 64 96 128 192 256; do
		echo "== base $b" >>"$out/enc.txt"
		ALPGPU_INIT_BASE=$b run 100 enc.txt python tools/r05_time_encode.py 1048576 mixed
	done
	unset ALPGPU_LIB
	grep "^== base\|^mixed" "$out/enc.txt" | cut -c1-200
	;;
52) run 200 hit.txt python tools/r05_hit_rate.py
	grep -v "amdgpu.ids\|^==" "$out/hit.txt"
	;;
final) # the closing run on the library as committed: whole GPU suite, smoke, the bench line, the configs[4] line at N = 1, the profile
	run 600 pytest.txt python -m pytest tests -m gpu -q
	tail -4 "$out/pytest.txt"
	run 120 smoke.txt python __graft_entry__.py smoke
	tail -1 "$out/smoke.txt"
	run 900 bench.txt python bench.py --steps 20 --warmup 5
	grep "^{" "$out/bench.txt" | tail -1 > "$out/bench.json"; wc -c "$out/bench.json"
	run 900 bench_configs4.txt python bench.py --gpus 1 --column-gb 100 --steps 10 --warmup 3
	grep "^{" "$out/bench_configs4.txt" | tail -1 > "$out/bench_configs4_n1.json"; wc -c "$out/bench_configs4_n1.json"
	timeout 1500 bash tools/profile_round.sh r05 > "$out/profile_round.txt" 2>&1
	echo "profile rc $?"
	;;
*)  echo "unknown step $step"; exit 2 ;;
esac
