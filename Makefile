.PHONY: install build test test-gpu lint typecheck notebook format format-check host-sanitize clean bench docs changelog release-prepare release

install:
	python -m pip install -e . --no-build-isolation --no-deps

build:
	python -m nanofed_amd.native_build

test: build
	python -m pytest tests -q -m "not gpu"

test-gpu: build
	python -u -m pytest tests -q -m gpu --timeout 120 --timeout-method thread

lint:
	python -m compileall -q nanofed_amd scripts tests tools bench.py
	@command -v ruff >/dev/null && ruff check nanofed_amd scripts tests tools bench.py || echo "ruff not installed: syntax check only"

typecheck:
	@python -c "import mypy" 2>/dev/null && python -m mypy nanofed_amd || echo "mypy not installed"

format:
	@command -v ruff >/dev/null && ruff format nanofed_amd tests || echo "ruff not installed"

format-check:
	@command -v ruff >/dev/null && ruff format --check nanofed_amd tests || echo "ruff not installed"

# AddressSanitizer + UBSan build of the host library and its C++ self-test
# (GPU sanitizers are not used: host code only)
host-sanitize:
	bash tools/host_sanitize.sh

bench: build
	python bench.py

clean:
	rm -rf build nanofed_amd/_lib/*.so .pytest_cache gpurun_out
	find . -name __pycache__ -type d -prune -exec rm -rf {} +

docs:
	@python -c "import sphinx" 2>/dev/null && python -m sphinx -b html docs/source docs/_build/html || echo "sphinx not installed: the documentation sources are the reStructuredText files under docs/source"

notebook:
	python tools/execute_notebook.py examples/mnist/tutorial.ipynb

changelog:
	python scripts/changelog.py $(VERSION)

release-prepare:
	python scripts/prepare_release.py $(VERSION)

release:
	bash scripts/release.sh
