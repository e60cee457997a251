"""Allan-deviation plugin -- same surface as the reference's demo_algorithms/allan_analysis.py
(input ['fs','accel','gyro'], output ['algo_time','ad_accel','ad_gyro']); the six axes go to the device as one
batch of six series.  Inside this package's Sim the plugin is handed the device-resident sensor series of ALL runs
(run_device) and no host copy of them is made."""
import numpy as np


class Allan(object):
    def __init__(self):
        self.input = ['fs', 'accel', 'gyro']
        self.output = ['algo_time', 'ad_accel', 'ad_gyro']
        self.batch = True
        self.results = None

    def run(self, set_of_input):
        import ginsim
        fs, accel, gyro = set_of_input[0], np.asarray(set_of_input[1]), np.asarray(set_of_input[2])
        series = np.concatenate([accel.T, gyro.T], axis=0)                   # (6, n)
        avar, tau = ginsim.allan_var_host(ginsim.default_context(), series, fs)
        self.results = [tau, np.sqrt(avar[0:3].T), np.sqrt(avar[3:6].T)]     # allan_analysis.py:47-49

    def run_device(self, sensor_job, fs):
        """Device protocol of this package's Sim: all runs of `sensor_job` (a ginsim.MonteCarloJob that kept its sensor
        series) at once.  Returns one [tau, ad_accel, ad_gyro] list per run, in run order."""
        tau, ad = sensor_job.allan(fs)
        per_run = [[tau, ad['accel'][r], ad['gyro'][r]] for r in range(sensor_job.runs)]
        self.results = per_run[-1]
        return per_run

    def get_results(self):
        return self.results

    def reset(self):
        pass
