"""Plugin ("algorithm") protocol.  Same duck-typed surface as the reference
(gnss_ins_sim/sim/ins_algo.py:10-67, README.md:173-248): attributes ``input``, ``output`` (lists of
registry names), optional ``name``, ``batch``, ``results``; methods ``run(set_of_input)``,
``get_results()``, ``reset()``.

Extension used only by this package's Sim: a plugin whose mechanisation exists inside the fused HIP kernel
advertises ``mc_algo`` ('free' | 'odo' | 'allan'); Sim then integrates ALL Monte-Carlo runs of that plugin in
one launch instead of calling ``run`` once per run.  Plugins without ``mc_algo`` are still hosted run by run.
"""


class InsAlgo(object):
    def __init__(self):
        self.input = []
        self.output = []
        self.batch = True
        self.results = None

    def run(self, set_of_input):
        raise NotImplementedError

    def update(self, set_of_input):
        raise NotImplementedError

    def get_results(self):
        return self.results

    def reset(self):
        pass
