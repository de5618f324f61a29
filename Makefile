# Convenience targets; the driver uses __graft_entry__.py / bench.py / pytest directly.
.PHONY: build oracle test-cpu test-gpu bench profile clean
build:
	python -m allocnet_amd.build
oracle:
	$(MAKE) -C oracle
test-cpu: build oracle
	python -m pytest tests -x -q -m "not gpu"
test-gpu:
	python -m pytest tests -x -q -m gpu
bench:
	python bench.py
profile:
	bash tools/profile.sh r01
clean:
	rm -f allocnet_amd/lib/*.so allocnet_amd/lib/*.o oracle/liboracle.so tests/cpp/test_facade
