"""Plugin host, interface-compatible with the reference's ``InsAlgoMgr``
(gnss_ins_sim/sim/ins_algo_manager.py:12-149): ``algo``, ``input``, ``output``, ``nin``, ``nout``, ``nalgo``,
``input_alloc``, ``output_alloc``, ``get_algo_name(i)`` and ``run_algo(input_data, keys)``.

``run_algo`` is the reference's algo x run loop and is only used for plugins that are NOT inside the fused
kernel (user code); the FreeIntegration plugins are batched by ``Sim`` (see ins_sim.py).
"""
import copy


class InsAlgoMgr(object):
    def __init__(self, algo):
        if algo is None:
            self.algo = None
        else:
            self.algo = algo if isinstance(algo, list) else [algo]
        self.input, self.output = [], []
        self.nin = self.nout = self.nalgo = 0
        self.input_alloc, self.output_alloc = [], []
        if self.algo is not None:
            self._check_algo()

    def _check_algo(self):
        try:
            for a in self.algo:
                if len(a.input) < 1 or len(a.output) < 1:
                    raise ValueError
        except Exception:
            raise ValueError('algorithm input or output is not a valid list or tuple.')
        # union, first-seen order (the reference uses set order, which is arbitrary; ins_algo_manager.py:135)
        for a in self.algo:
            for name in a.input:
                if name not in self.input:
                    self.input.append(name)
            for name in a.output:
                if name not in self.output:
                    self.output.append(name)
        for a in self.algo:
            self.input_alloc.append([self.input.index(i) for i in a.input])
            self.output_alloc.append([self.output.index(i) for i in a.output])
        self.nin, self.nout, self.nalgo = len(self.input), len(self.output), len(self.algo)

    def get_algo_name(self, i):
        if self.algo is None or i >= self.nalgo:
            return None
        return getattr(self.algo[i], 'name', 'algo' + str(i))

    def run_algo(self, input_data, keys=None, only=None):
        """Reference loop (ins_algo_manager.py:39-96): for each algorithm, for each key: reset(), run(deep copy of
        its inputs), get_results(); results keyed '<algo name>_<key>'.  ``only`` restricts to some algorithm indices."""
        if len(input_data) != self.nin:
            raise ValueError('Required %s input, but provide %s.' % (self.nin, len(input_data)))
        results = [{} for _ in range(self.nout)]
        if keys is None:
            keys = [0]
            for d in input_data:
                if hasattr(d, 'keys'):
                    keys = list(d.keys())
                    break
        for i in range(self.nalgo):
            if only is not None and i not in only:
                continue
            name = self.get_algo_name(i)
            for key in keys:
                self.algo[i].reset()
                args = []
                for j in self.input_alloc[i]:
                    d = input_data[j]
                    if hasattr(d, 'keys'):
                        if key not in d:
                            raise ValueError("set_of_input has keys %s, but you are requiring %s" % (list(d.keys())[:8], key))
                        args.append(d[key])
                    else:
                        args.append(d)
                self.algo[i].run(copy.deepcopy(args))
                out = self.algo[i].get_results()
                for j, slot in enumerate(self.output_alloc[i]):
                    results[slot][name + '_' + str(key)] = out[j]
        return results
