"""MI355X-native drop-in for the reference package `galaxify` (src/galaxify/__init__.py is empty
there too): `from galaxify import simulation` keeps working, backed by HIP kernels."""
