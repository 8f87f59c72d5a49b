# Convenience targets (everything is plain Python / hipcc underneath; see README.md).
.PHONY: build test test-gpu bench profile clean
build:            ## hipcc --offload-arch=gfx950 -> mpopis_amd/lib/libmpopis_hip.so, gcc -> oracle/libmpopis_oracle.so (the checker)
	python -c "import __graft_entry__ as g; g.build()"
test: build       ## CPU suite: oracle KATs, ABI (incl. the C99 client), distributed (gloo)
	python -m pytest tests -q -m "not gpu"
test-gpu: build   ## parity suite on an MI355X
	python -m pytest tests -q -m gpu
bench: build      ## one JSON line (roofline + cpu_baseline)
	python bench.py
profile:          ## rocprofv3 evidence for profiles/ (on the GPU box)
	bash tools/profile_round.sh r02 && bash tools/prof_c4.sh r02_c4 8
clean:
	rm -rf mpopis_amd/lib/obj mpopis_amd/lib/*.so oracle/*.so tools/*_bin tests/shim/*.so
