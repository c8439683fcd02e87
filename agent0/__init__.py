"""Drop-in module path: ``agent0.deepq.*`` / ``agent0.common.*`` resolve to the MI355X-native implementation in
``agent0_amd`` so that ``python -m agent0.deepq.main`` and ``python -m agent0.deepq.launch`` keep working (reference
README.md:34-53)."""
