"""Sphinx configuration of the nanofed-amd documentation."""

import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]

project = "nanofed-amd"
author = "nanofed-amd developers"
copyright = "2026, nanofed-amd developers"  # noqa: A001
release = re.search(
    r'(?m)^version\s*=\s*"([^"]+)"', (ROOT / "pyproject.toml").read_text()
).group(1)
version = ".".join(release.split(".")[:2])

extensions = [
    "sphinx.ext.autodoc",
    "sphinx.ext.napoleon",
    "sphinx.ext.viewcode",
    "sphinx.ext.intersphinx",
]
autodoc_mock_imports = ["torch", "aiohttp", "numpy", "pydantic"]
autodoc_member_order = "bysource"
intersphinx_mapping = {
    "python": ("https://docs.python.org/3", None),
    "torch": ("https://pytorch.org/docs/stable", None),
}

templates_path = ["_templates"]
exclude_patterns = ["release_notes/template.rst"]
html_theme = "furo"
html_static_path = ["_static"]
html_title = f"nanofed-amd {release}"
pygments_style = "default"
nitpicky = False
