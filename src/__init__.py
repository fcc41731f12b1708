"""Drop-in import shim: the reference's scripts import `src.models.*` / `src.pipelines.*`
(`scripts/pose2vid.py:21-30`, `scripts/audio2vid.py:22-35`).  With this repository first on `sys.path`
those imports resolve to the MI355X-native implementations in `aniportrait_amd`."""
