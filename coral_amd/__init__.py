"""coral_amd — MI355X-native breakpoint-graph construction for CoRAL `reconstruct`.

Only the hot path of SURVEY.md §8 lives here: the HIP kernels + C-ABI under
``csrc/`` and the host-side mirror of the reference's operator interface
(``infer_breakpoint_graph.reconstruct_graph`` and the object it returns).
"""

__version__ = "0.1.0"
