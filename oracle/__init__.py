"""CPU oracle for the honk2 keyword-spotting inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``honk2_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker, never as the thing
that is measured or shipped.

Parity status
-------------
* models (``oracle.models``): PINNED.  The restatement is checked against the
  reference's own ``model.ResNet`` / ``model.CNN`` classes (imported from
  /root/reference by ``oracle/gen_golden.py`` in the build container) through
  the fixtures committed under ``tests/golden/``.
* front end (``oracle.frontend``): PARITY UNPINNED for the librosa part.  The
  arithmetic of ``utils/audio_processor.py:19-26`` lives in ``librosa``
  (unpinned in ``requirements.txt:7``, must be < 0.10, not installed, not
  vendored, no network).  The oracle restates librosa's published algorithm
  (SURVEY.md Appendix A); the ``scipy.fftpack.dct`` step of ``:28`` IS pinned
  (scipy is installed; see ``tests/golden/frontend_dct_pin.npz``).
"""
