"""RadarOperator with the public interface of the reference's driver class (radar_operator.py:35-425).

Where the reference forks a process pool and maps one worker call per ray (:231-256), this class hands
the whole list of rays of a scan to the GPU in one cpol_run_rays call.  With torch.distributed
initialised (one process per GPU) the rays are sharded over the ranks and the volume is gathered on
rank 0 (parallel.py)."""
import numpy as np

from . import cfg, model as model_mod, runtime
from ._lib import OBS_NAMES
from .constants import constants
from .radar_output import SimulatedRadar

BASE_VARIABLES = ['U', 'V', 'W', 'QR_v', 'QS_v', 'QG_v', 'RHO', 'T']
BASE_VARIABLES_2MOM = ['QH_v', 'QNH_v', 'QNR_v', 'QNS_v', 'QNG_v']


class RadarOperator(object):
    def __init__(self, options_file='', diagnostic_mode=False, device=0):
        if isinstance(options_file, dict):
            cfg.set_config(options_file)
        else:
            cfg.init(options_file)
        constants.update()
        self.config = cfg
        self.lut_sz = None
        self.current_microphys_scheme = 1
        self.dic_vars = {}
        self.N = 0
        self.diagnostic_mode = diagnostic_mode
        self.device = device

    # -- configuration / state -----------------------------------------------------------------
    def get_config(self):
        return cfg.CONFIG

    def set_config(self, options_file):
        if isinstance(options_file, dict):
            cfg.set_config(options_file)
        else:
            cfg.init(options_file)
        constants.update()

    def set_lut(self, hydro_scheme, lut=None):
        """lut: {class letter: Lookup_table}; without it the tables of the configured frequency are read from
        read_lut.FOLDER_LUT like lookup/read_lut.py:10-20 does (flat .clut files, see read_lut)."""
        if lut is None:
            from . import read_lut
            lut = read_lut.get_lookup_tables(hydro_scheme, cfg.CONFIG['radar']['frequency'])
        self.lut_sz = lut

    def get_pos_and_time(self):
        c = cfg.CONFIG['radar']['coords']
        t = self.dic_vars['T'].attributes.get('time', 0) if self.dic_vars else 0
        return {'latitude': c[0], 'longitude': c[1], 'altitude': c[2], 'time': t}

    def load_model_file(self, filename, cfilename=''):
        """filename: path of an .npz written by cosmo_pol_amd.model.save_npz, of a NetCDF file of the model (.nc,
        classic format), or an already built {name: ModelVariable} dictionary.  The microphysics scheme follows
        the variables present (radar_operator.py:124-132); the refractivity is derived when refraction scheme 2
        is configured (:94-96)."""
        if isinstance(filename, dict):
            dic = filename
        elif str(filename).endswith(('.nc', '.nc3', '.cdf')):
            dic = model_mod.load_netcdf(filename, with_refractivity=cfg.CONFIG['refraction']['scheme'] == 2)
        else:
            dic = model_mod.load_npz(filename)
        missing = [v for v in BASE_VARIABLES if v not in dic]
        if missing:
            raise ValueError('Not all necessary variables could be found in file: missing %s' % missing)
        two_mom = all(v in dic for v in BASE_VARIABLES_2MOM)
        names = BASE_VARIABLES + (BASE_VARIABLES_2MOM if two_mom else [])
        if 'EDR' in dic:                 # eddy dissipation rate: turbulence broadening of the spectrum (doppler_sz.py:36)
            names = names + ['EDR']
        self.dic_vars = {k: dic[k] for k in names}
        self.N = dic['N'] if 'N' in dic else 0          # refractivity, kept apart (radar_operator.py:141-143)
        cfg.CONFIG['microphysics']['scheme'] = 2 if two_mom else 1
        self.current_microphys_scheme = cfg.CONFIG['microphysics']['scheme']

    # -- scans ---------------------------------------------------------------------------------
    def _run(self, scan_type, azs, els, n_sweeps, rays_per_sweep, sweep_elev, sweep_az):
        if not self.dic_vars:
            print('No model file has been loaded! Aborting...')
            return None
        if cfg.CONFIG['radar']['type'] != 'ground':
            print('PPI / RHI profiles only possible for ground radars')
            return []
        if self.lut_sz is None:
            raise RuntimeError('lookup tables not set: call set_lut(scheme, lut) first')
        dscheme = cfg.CONFIG['doppler']['scheme']
        if dscheme not in (1, 2, 3, 4):
            raise ValueError('doppler.scheme must be 1, 2, 3 or 4, got %r' % (dscheme,))
        st = runtime.get_engine(self.device, doppler=dscheme, sensitivity_cut=True)
        runtime.ensure_luts(st, self.lut_sz)
        runtime.ensure_model(st, self.dic_vars)
        runtime.ensure_quadrature(st)
        ctx = st.eng.ctx
        if cfg.CONFIG['refraction']['scheme'] == 2 and self.N != 0:
            st.eng.set_refraction(self.N, np.unique(els))
        else:
            if cfg.CONFIG['refraction']['scheme'] == 2:
                print('No refractivity field N in the model state. Using 4/3 method instead.')   # radar_operator.py:99-102
            st.eng.set_refraction(None)
        rank, world = _dist_info()
        n_total = len(azs)
        if world == 1:
            out = ctx.run_rays(azs, els, diag=self.diagnostic_mode)      # raises IndexError when a gate leaves the model domain
            fields = {n: out['obs'][i] for i, n in enumerate(OBS_NAMES)}
            fields['RVEL'] = out['rvel']
            if dscheme in (3, 4):
                fields['DSPECTRUM'] = out['dspectrum']              # [n_rays, n_gates, n_velocity_bins]
            extra = {'mask': out['mask'], 'lat': out['lat'], 'lon': out['lon'], 'height': out['height']}
            if self.diagnostic_mode:
                # radar_operator.py:245-246: the model variables averaged over the sub-beams join the beam's fields
                fields.update({n: out['diag'][i] for i, n in enumerate(ctx.var_names)})
        else:
            # one process per GPU: every rank simulates its share of the rays (the reference's pool.map over rays,
            # radar_operator.py:253-256) and rank 0 receives the volume in one gather
            vol = self._run_sharded(ctx, azs, els, n_sweeps, rays_per_sweep, rank, world)
            if vol is None:
                return None
            fields = {n: vol['obs'][:, i] for i, n in enumerate(OBS_NAMES)}
            fields['RVEL'] = vol['rvel']
            if dscheme in (3, 4):
                fields['DSPECTRUM'] = vol['dspectrum']
            extra = {k: vol[k] for k in ('mask', 'lat', 'lon', 'height')}
            if self.diagnostic_mode:
                fields.update({n: vol['diag'][:, i] for i, n in enumerate(ctx.var_names)})
        ranges = st.eng.n_gates and np.arange(cfg.CONFIG['radar']['radial_resolution'] / 2,
                                              cfg.CONFIG['radar']['range'], cfg.CONFIG['radar']['radial_resolution'])
        return SimulatedRadar(scan_type, els, azs, ranges, fields, extra['mask'], extra['lat'], extra['lon'],
                              extra['height'], n_sweeps, rays_per_sweep, self.get_pos_and_time(),
                              fixed_angles=sweep_elev if scan_type == 'ppi' else sweep_az,
                              varray=constants.VARRAY)

    def _run_sharded(self, ctx, azs, els, n_sweeps, rays_per_sweep, rank, world):
        """Rays block-cyclic over the ranks, one gather of every output field to rank 0.  With the RCCL backend
        the outputs stay in device memory until rank 0 has the whole volume; with gloo (CPU tensors) they go
        through the host.  Returns {name: numpy array [n_rays, ...]} on rank 0, None elsewhere."""
        from collections import OrderedDict
        import torch
        import torch.distributed as dist
        from . import parallel
        from ._lib import ERR_DOMAIN, ERR_REFRACTION, CosmoPolError
        shards = parallel.all_shards(n_sweeps, rays_per_sweep, world)
        idx = shards[rank]
        if dist.get_backend() == 'nccl':
            dev = torch.device('cuda', self.device)
            local, status = parallel.run_rays_device(ctx, azs[idx], els[idx], dev, diag=self.diagnostic_mode)
            status = status.cpu().numpy()
        else:
            dev = None
            out = ctx.run_rays(azs[idx], els[idx], check_status=False, diag=self.diagnostic_mode)
            local = OrderedDict((k, torch.as_tensor(out[k])) for k in ('obs', 'rvel', 'mask', 'lat', 'lon', 'dist', 'height', 'dspectrum', 'diag')
                                if k in out)
            if 'diag' in local:
                local['diag'] = local['diag'].permute(1, 0, 2)
            if 'rvel' not in local:
                local['rvel'] = torch.full(tuple(local['mask'].shape), float('nan'), dtype=torch.float64)
            local['obs'] = local['obs'].permute(1, 0, 2)
            status = out['ray_status']
        # every rank learns of a failed ray anywhere, so that all of them raise instead of waiting in the gather
        if parallel.any_rank((status == ERR_DOMAIN).any(), dev):
            raise IndexError('ERROR: RADAR DOMAIN IS NOT ENTIRELY CONTAINED IN COSMO SIMULATION DOMAIN: ABORTING')
        if parallel.any_rank((status == ERR_REFRACTION).any(), dev):
            raise CosmoPolError('cpol_run_rays: the refraction table has no profile for the elevation of some rays')
        vol = parallel.gather_volume(local, shards, len(azs))
        if vol is None:
            return None
        return {k: v.cpu().numpy() for k, v in vol.items()}

    def get_GPM_swath(self, GPM_file, band='Ku'):
        """radar_operator.py:350-424: the swath of a GPM DPR overpass.  GPM_file: the DPR HDF5 file (needs h5py) or
        a dictionary with the arrays gpm.get_GPM_angles reads.  Returns a gpm.SimulatedGPM."""
        from . import gpm
        if not self.dic_vars:
            print('No model file has been loaded! Aborting...')
            return None
        if self.lut_sz is None:
            raise RuntimeError('lookup tables not set: call set_lut(scheme, lut) first')
        if band == 'Ku':
            cfg.CONFIG['radar']['frequency'] = 13.6
        elif band == 'Ka':
            cfg.CONFIG['radar']['frequency'] = 35.6
        cfg.CONFIG['radar']['sensitivity'] = constants.GPM_SENSITIVITY
        cfg.CONFIG['radar']['type'] = 'GPM'
        cfg.CONFIG['radar']['radial_resolution'] = constants.GPM_RADIAL_RES_KA if band == 'Ka' \
            else constants.GPM_RADIAL_RES_KU
        az, elev, rang, coords_GPM = gpm.get_GPM_angles(GPM_file, band)
        st = runtime.get_engine(self.device, doppler=0, sensitivity_cut=True)
        runtime.ensure_luts(st, self.lut_sz)
        runtime.ensure_model(st, self.dic_vars)
        beams = gpm.simulate_swath(st.eng, cfg.CONFIG, az, elev, rang, coords_GPM, band)
        st.conf_snapshot = None              # the engine was reconfigured per scan line
        st.quad_key = None
        return gpm.SimulatedGPM(beams, az.shape, band)

    def get_PPI(self, elevations, azimuths=[], az_step=-1, az_start=0, az_stop=359):
        """radar_operator.py:195-273."""
        if np.isscalar(elevations):
            elevations = [elevations]
        if az_step == -1:
            az_step = cfg.CONFIG['radar']['3dB_beamwidth']
        if not len(azimuths):
            if az_start > az_stop:
                azimuths = np.hstack((np.arange(az_start, 360., az_step), np.arange(0, az_stop + az_step, az_step)))
            else:
                azimuths = np.arange(az_start, az_stop + az_step, az_step)
        azimuths = np.asarray(azimuths, dtype=np.float64)
        azs = np.tile(azimuths, len(elevations))
        els = np.repeat(np.asarray(elevations, dtype=np.float64), len(azimuths))
        return self._run('ppi', azs, els, len(elevations), len(azimuths), elevations, azimuths)

    def get_RHI(self, azimuths, elevations=[], elev_step=-1, elev_start=0, elev_stop=90):
        """radar_operator.py:275-348."""
        if np.isscalar(azimuths):
            azimuths = [azimuths]
        if not len(elevations):
            if elev_step == -1:
                elev_step = cfg.CONFIG['radar']['3dB_beamwidth']
            elevations = np.arange(elev_start, elev_stop + elev_step, elev_step)
        elevations = np.asarray(elevations, dtype=np.float64)
        azs = np.repeat(np.asarray(azimuths, dtype=np.float64), len(elevations))
        els = np.tile(elevations, len(azimuths))
        return self._run('rhi', azs, els, len(azimuths), len(elevations), elevations, azimuths)

    def get_VPROF(self):
        """radar_operator.py:153-192: one vertical ray."""
        return self._run('vprof', np.array([0.]), np.array([90.]), 1, 1, [90.], [0.])


def _dist_info():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1
